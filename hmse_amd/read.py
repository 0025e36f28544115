"""The read path on the GPU: stored records -> original bytes, verified by SHA-256.

Three branches as in the reference (README.md:1621-1675, 2191-2198): FULL -> inflate; DELTA -> inflate with the
base chunk as dictionary; POINTER -> the target chunk's bytes.  All of it runs in the hand-written kernels of
hmse_amd/csrc/l1_inflate.hip and the L3 SHA-256 kernel (the reference's "100 % checksum pass" gate, README.md:1329).
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from .config import KIND_DELTA, KIND_POINTER
from .manifest import Manifest, Store


class ReadError(RuntimeError):
    pass


def reconstruct_shard(res, verify: bool = True) -> torch.Tensor:
    """Inverse of ingest_shard for a ShardResult whose first occurrences are all local (one shard)."""
    cuts = res.cuts
    n_chunks = cuts.numel() - 1
    dev = cuts.device
    lens = cuts[1:] - cuts[:-1]
    if res.streams is None:
        raise ReadError("reconstruct_shard needs the L1 layer's streams")
    if getattr(res, "base_global", None) is not None and bool(((res.base_global >= 0) & (res.base < 0)).any()):
        raise ReadError("records of this shard use dictionaries stored on other shards: read.reconstruct_shards decodes all shards together")
    raw, raw_off, _ = ops.l1_inflate(res.streams, res.stream_off, res.kind, res.base, lens[res.uniq_ids])
    slot_of = torch.full((n_chunks,), -1, dtype=torch.int64, device=dev)
    slot_of[res.uniq_ids] = torch.arange(res.uniq_ids.numel(), dtype=torch.int64, device=dev)
    if res.first_occ is not None:
        fo = res.first_occ - res.chunk_base
        if bool(((fo < 0) | (fo >= n_chunks)).any()):
            raise ReadError("a first occurrence lives on another shard")
        slot_of = slot_of[fo]
    data = ops.read_assemble(cuts, slot_of, raw_off, raw)
    if verify and res.digests is not None:
        verify_digests(data, cuts, res.digests)
    return data


def reconstruct_shards(results: list, verify: bool = True) -> list:
    """Inverse of a sharded ingest whose records may use dictionaries stored on OTHER shards (ingest_shard(global_l4=True)),
    or of a multi-rank STREAM (stream_dist.DistStreamIngest: the ranks' chunks interleave in global order, `res.gidx` names
    every local chunk's global index and a first occurrence may live on any rank):
    the stored records of all shards are inflated in ONE call, in global stored-chunk order (= rank order), so that a
    DELTA record's dictionary — named by its global stored-chunk index — is any earlier record; then every shard's chunks
    are laid out from the slots their first occurrences name.  Returns the shards' data tensors, in order."""
    dev = results[0].cuts.device
    if all(getattr(r, "ug", None) is not None for r in results):
        return _reconstruct_stream_global_l4(results, verify)
    u_counts = [int(r.uniq_ids.numel()) for r in results]
    u_bases = [sum(u_counts[:i]) for i in range(len(results))]
    lens_u, offs, run = [], [], 0
    for r in results:
        if r.streams is None:
            raise ReadError("reconstruct_shards needs the L1 layer's streams")
        ln = r.cuts[1:] - r.cuts[:-1]
        lens_u.append(ln[r.uniq_ids]); offs.append(r.stream_off[:-1] + run); run += int(r.streams.numel())
    streams = torch.cat([r.streams for r in results])
    stream_off = torch.cat(offs + [torch.tensor([run], dtype=torch.int64, device=dev)])
    kind = torch.cat([r.kind for r in results])
    base = torch.cat([r.base_global if r.base_global is not None else torch.where(r.base >= 0, r.base + ub, r.base)
                      for r, ub in zip(results, u_bases)])
    raw, raw_off, _ = ops.l1_inflate(streams, stream_off, kind, base, torch.cat(lens_u))
    # global chunk index -> global stored-chunk slot (first occurrences only)
    n_global = results[0].n_global
    slot_of_global = torch.full((n_global,), -1, dtype=torch.int64, device=dev)
    for r, ub in zip(results, u_bases):
        g = getattr(r, "gidx", None)
        own = g[r.uniq_ids] if g is not None else r.chunk_base + r.uniq_ids
        slot_of_global[own] = torch.arange(ub, ub + r.uniq_ids.numel(), dtype=torch.int64, device=dev)
    out = []
    for r in results:
        n_chunks = r.cuts.numel() - 1
        fo = r.first_occ if r.first_occ is not None else torch.arange(r.chunk_base, r.chunk_base + n_chunks, dtype=torch.int64, device=dev)
        if bool(((fo < 0) | (fo >= n_global)).any()):
            raise ReadError("a first occurrence lies outside the global chunk range")
        slots = slot_of_global[fo]
        if bool((slots < 0).any()):
            raise ReadError("a first occurrence is not a stored chunk of any shard")
        d = ops.read_assemble(r.cuts, slots, raw_off, raw)
        if verify and r.digests is not None:
            verify_digests(d, r.cuts, r.digests)
        out.append(d)
    return out


def _reconstruct_stream_global_l4(results: list, verify: bool) -> list:
    """reconstruct_shards for the ranks of a global-L4 STREAM (stream_dist.GlobalL4StreamIngest): `res.ug` names every stored
    chunk's GLOBAL stored index in stream order and `res.base_global` a dictionary by that index — which may belong to any rank,
    also a later-numbered one — so the records of all ranks are decoded in that order (a dictionary then always precedes)."""
    dev = results[0].cuts.device
    starts, slens, lens_u, run = [], [], [], 0
    for r in results:
        ln = r.cuts[1:] - r.cuts[:-1]
        lens_u.append(ln[r.uniq_ids]); starts.append(r.stream_off[:-1] + run); slens.append(r.stream_off[1:] - r.stream_off[:-1])
        run += int(r.streams.numel())
    ug = torch.cat([r.ug for r in results])
    u = ug.numel()
    order = torch.argsort(ug)
    if not torch.equal(ug[order], torch.arange(u, dtype=torch.int64, device=dev)):
        raise ReadError("the ranks' stored chunks do not tile the global stored order")
    pick = lambda parts: torch.cat(parts)[order]
    raw, raw_off, _ = ops.l1_inflate(torch.cat([r.streams for r in results]), pick(starts), pick([r.kind for r in results]),
                                     pick([r.base_global for r in results]), pick(lens_u), stream_len=pick(slens).to(torch.int32))
    n_global = results[0].n_global
    slot_of_global = torch.full((n_global,), -1, dtype=torch.int64, device=dev)
    for r in results:
        slot_of_global[r.gidx[r.uniq_ids]] = r.ug
    out = []
    for r in results:
        slots = slot_of_global[r.first_occ]
        if bool((slots < 0).any()):
            raise ReadError("a first occurrence is not a stored chunk of any rank")
        d = ops.read_assemble(r.cuts, slots, raw_off, raw)
        if verify and r.digests is not None:
            verify_digests(d, r.cuts, r.digests)
        out.append(d)
    return out


def verify_stored(res) -> int:
    """Decode every record this shard stores (FULL and DELTA) and re-check its SHA-256 against the L3 digest: the
    reference's read-side gate (README.md:1329) without the POINTER branch, so it holds on any rank of a sharded run
    (a pointer's target may live on another GPU).  Returns the number of records verified."""
    if res.streams is None or res.digests is None:
        raise ReadError("verify_stored needs the L1 streams and the L3 digests")
    lens = res.cuts[1:] - res.cuts[:-1]
    raw, raw_off, _ = ops.l1_inflate(res.streams, res.stream_off, res.kind, res.base, lens[res.uniq_ids])
    verify_digests(raw, raw_off, res.digests[res.uniq_ids])
    return int(res.uniq_ids.numel())


def verify_digests(data: torch.Tensor, cuts: torch.Tensor, digests: torch.Tensor) -> None:
    got = ops.l3_sha256(data, cuts)
    if not torch.equal(got, digests):
        bad = int((got != digests).any(dim=1).sum().item())
        raise ReadError(f"SHA-256 mismatch on {bad} of {digests.shape[0]} reconstructed chunks")


def parse_manifest(m: Manifest) -> dict:
    """Host-side decode of the record headers: per stored slot its kind, dictionary slot, stream position inside the blob and raw length."""
    idx, cmap = m.index, m.chunk_map
    u = len(idx)
    own = cmap["kind"] != KIND_POINTER
    slot_kind = np.zeros(u, np.uint8)
    slot_kind[cmap["slot"][own]] = cmap["kind"][own]
    raw_len = np.zeros(u, np.int64)
    raw_len[cmap["slot"][own]] = cmap["raw_length"][own]      # every stored slot is some local chunk's own record
    rec_off = idx["lba"].astype(np.int64) * m.lba_unit
    is_delta = slot_kind == KIND_DELTA
    s_off = rec_off + np.where(is_delta, 8, 0)
    s_len = idx["length"].astype(np.int64) - np.where(is_delta, 8, 0)
    base = np.full(u, -1, np.int64)
    base_shard = np.full(u, m.shard, np.int64)          # which shard's slot `base` names (global L4: possibly an earlier shard's)
    if is_delta.any():
        # DeltaChunk header {base_lba u32, base_length u16, delta_length u16} (README.md:2182-2189)
        hdr_pos = rec_off[is_delta][:, None] + np.arange(8)[None, :]
        hdr = m.blob[hdr_pos].copy().view("<u4")  # [:,0] base_lba, [:,1] lengths
        s_len[is_delta] = hdr[:, 1] >> 16
        local = np.ones(u, bool)
        if m.n_remote():                                 # dictionaries in another shard's blob: named by the manifest's table
            rb = m.remote_bases
            local[rb["slot"]] = False
            base[rb["slot"]] = rb["base_slot"]; base_shard[rb["slot"]] = rb["shard"]
        loc = local[is_delta]
        order = np.argsort(idx["lba"], kind="stable")
        pos = np.searchsorted(idx["lba"][order], hdr[loc, 0])
        if (pos >= u).any() or (idx["lba"][order][np.minimum(pos, u - 1)] != hdr[loc, 0]).any():
            raise ReadError("a DeltaChunk header names an LBA that is not in the index")
        base[np.nonzero(is_delta)[0][loc]] = order[pos]
    return {"kind": slot_kind, "base": base, "base_shard": base_shard, "stream_off": s_off, "stream_len": s_len, "raw_len": raw_len}


MAX_DELTA_DEPTH_LOG2 = 16     # a dictionary chain of more than 65536 records is refused as corrupt (read.dependency_order)


def dependency_order(base: np.ndarray):
    """Records in (shard, slot) order with `base[k]` = the record holding k's dictionary (-1 none).  hmse_l1_inflate wants a
    dictionary to PRECEDE its dependants; that holds as stored except in the store of a multi-rank stream ingested with global L4,
    where a dictionary may sit on a later-numbered shard.  -> None if base[k] < k everywhere, else (order, new_of_old): records
    sorted by dictionary depth (stable), and the inverse map."""
    n = len(base)
    if n == 0 or (base < np.arange(n)).all():
        return None
    # depth of every record in the dictionary forest by POINTER DOUBLING: after round r, `anc` is the 2^r-th ancestor (n = past the
    # root) and `depth` counts the links walked so far — ceil(log2(longest chain)) + 1 numpy passes instead of one pass per level
    # (a long version chain of a global-L4 stream store cost O(n * depth); a corrupt store with a cycle O(n^2) before it was refused).
    anc = np.where(base >= 0, base, n).astype(np.int64)
    anc = np.concatenate([anc, np.array([n], np.int64)])      # the sentinel is its own ancestor
    depth = (anc[:n] < n).astype(np.int64)
    for _ in range(MAX_DELTA_DEPTH_LOG2 + 1):
        live = anc[:n] < n
        if not live.any():
            break
        depth = depth + np.where(live, np.concatenate([depth, [0]])[anc[:n]], 0)
        anc[:n] = anc[anc[:n]]
    else:
        raise ReadError(f"the DELTA records' dictionaries form a cycle (or a chain longer than {1 << MAX_DELTA_DEPTH_LOG2} records)")
    order = np.argsort(depth, kind="stable")
    new_of_old = np.empty(n, np.int64)
    new_of_old[order] = np.arange(n)
    return order, new_of_old


def read_store(store: Store, device, verify: bool = True) -> torch.Tensor:
    """A sharded store (one Manifest per shard, cross-shard pointers resolved by manifest.merge_manifests) -> the
    original corpus in global chunk order, decoded on `device`: the records of all shards are inflated in one call, then
    ONE assembly pass lays out every chunk from the slot its map entry names — its own shard's or, for a cross-shard
    POINTER, another's (README.md:1635-1669)."""
    from .manifest import PTR_UNRESOLVED
    if any(((m.pointers["flags"] & PTR_UNRESOLVED) != 0).any() for m in store.shards):
        raise ReadError("the store has unresolved cross-shard pointers: merge_manifests() its shards first")
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).copy()).to(dt).to(device)
    # ONE inflate call over the records of all shards, in (shard, slot) order: a DELTA's dictionary is an earlier record of
    # its own shard or — in a store ingested with global L4 — of an earlier shard (manifest.remote_bases)
    sb = np.cumsum([0] + [len(m.index) for m in store.shards])
    bb = np.cumsum([0] + [int(m.blob.size) for m in store.shards])
    ps = [parse_manifest(m) for m in store.shards]
    for m, p in zip(store.shards, ps):
        if m.n_remote():
            rb = m.remote_bases
            lba = m.blob[(m.index["lba"][rb["slot"]].astype(np.int64) * m.lba_unit)[:, None] + np.arange(4)[None, :]].copy().view("<u4")[:, 0]
            want = np.array([store.shards[int(r)].index["lba"][int(b)] for r, b in zip(rb["shard"], rb["base_slot"])], np.uint32)
            if not np.array_equal(lba, want):
                raise ReadError("the store has unresolved cross-shard DeltaChunk headers: merge_manifests() its shards first")
    cat = lambda key, adj=None: np.concatenate([(p[key] if adj is None else adj(i, p)) for i, p in enumerate(ps)]) if ps else np.zeros(0, np.int64)
    base_g = cat("base", lambda i, p: np.where(p["base"] >= 0, sb[p["base_shard"]] + p["base"], -1)).astype(np.int64)
    dep = dependency_order(base_g)          # (a global-L4 stream's store: dictionaries on later-numbered shards)
    o = (lambda a: a) if dep is None else (lambda a: a[dep[0]])
    if dep is not None:
        base_g = np.where(base_g >= 0, dep[1][np.maximum(base_g, 0)], -1)
    if len(base_g):
        blobs = [t(m.blob, torch.uint8) for m in store.shards if m.blob.size]
        blob_all = blobs[0] if len(blobs) == 1 else torch.cat(blobs) if blobs else torch.zeros(1, dtype=torch.uint8, device=device)
        del blobs
        raw_all, raw_off_all, _ = ops.l1_inflate(blob_all,
                                                 t(o(cat("stream_off", lambda i, p: p["stream_off"] + bb[i])), torch.int64),
                                                 t(o(cat("kind")), torch.uint8), t(o(base_g), torch.int64), t(o(cat("raw_len")), torch.int64),
                                                 stream_len=t(o(cat("stream_len")), torch.int32))
    else:
        raw_all = torch.empty(0, dtype=torch.uint8, device=device); raw_off_all = torch.zeros(1, dtype=torch.int64, device=device)
    slot_g = np.concatenate([sb[m.chunk_map["shard"].astype(np.int64)] + m.chunk_map["slot"].astype(np.int64) for m in store.shards]) \
        if store.shards else np.zeros(0, np.int64)
    lens = np.concatenate([m.chunk_map["raw_length"].astype(np.int64) for m in store.shards]) if store.shards else np.zeros(0, np.int64)
    from .manifest import stream_order
    perm = stream_order(store.shards)          # a multi-rank stream's store: the original bytes are the chunks in stream order
    if perm is not None:
        slot_g, lens = slot_g[perm], lens[perm]
    cuts = torch.zeros(len(lens) + 1, dtype=torch.int64, device=device)
    torch.cumsum(t(lens, torch.int64), 0, out=cuts[1:])
    data = ops.read_assemble(cuts, t(slot_g if dep is None else dep[1][slot_g], torch.int64), raw_off_all, raw_all)
    if verify:
        sha = np.concatenate([m.index["sha256"] for m in store.shards]) if store.shards else np.zeros((0, 32), np.uint8)
        if len(sha) and sha.any():
            verify_digests(data, cuts, t(sha[slot_g], torch.uint8))
    return data


def read_manifest(m: Manifest, device, verify: bool = True) -> torch.Tensor:
    """Manifest bytes (hmse_amd/manifest.py record formats) -> original data, decoded on `device`."""
    idx, cmap = m.index, m.chunk_map
    u, n = len(idx), len(cmap)
    if m.n_remote():
        raise ReadError("records of this manifest use dictionaries stored in other shards: read_store() the merged store")
    p = parse_manifest(m)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).copy()).to(dt).to(device)
    raw, raw_off, _ = ops.l1_inflate(t(m.blob, torch.uint8), t(p["stream_off"], torch.int64), t(p["kind"], torch.uint8), t(p["base"], torch.int64),
                                     t(p["raw_len"], torch.int64), stream_len=t(p["stream_len"], torch.int32))
    cuts = torch.zeros(n + 1, dtype=torch.int64, device=device)
    torch.cumsum(t(p["raw_len"][cmap["slot"]], torch.int64), 0, out=cuts[1:])
    data = ops.read_assemble(cuts, t(cmap["slot"].astype(np.int64), torch.int64), raw_off, raw)
    if verify and u and idx["sha256"].any():
        verify_digests(data, cuts, t(idx["sha256"][cmap["slot"]], torch.uint8))
    return data


class StoreReader:
    """Request-driven read (README.md:1444-1448 "Read Request (offset, len)" -> chunk map; 1621-1675 the three branches per
    requested chunk; gate README.md:1329 "1000 random articles"): a store (or one manifest) is opened ONCE — blobs to HBM, record
    headers parsed — and then serves byte ranges of the original corpus: chunk map -> the touched chunks -> their stored slots
    (a POINTER names its target's) -> the transitive closure over DELTA dictionaries -> ONE hmse_l1_inflate over that subset
    (dictionaries renumbered so that base < k) -> only the requested bytes are laid out -> SHA-256 of every touched chunk.
    Nothing outside the closure is decoded."""

    def __init__(self, store, device):
        from .manifest import PTR_UNRESOLVED
        shards = store.shards if isinstance(store, Store) else [store]
        if any(((m.pointers["flags"] & PTR_UNRESOLVED) != 0).any() for m in shards):
            raise ReadError("the store has unresolved cross-shard pointers: merge_manifests() its shards first")
        self.dev = device
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).copy()).to(dt).to(device)
        sb = np.cumsum([0] + [len(m.index) for m in shards])
        bb = np.cumsum([0] + [int(m.blob.size) for m in shards])
        ps = [parse_manifest(m) for m in shards]
        cat = lambda key, adj=None: np.concatenate([(p[key] if adj is None else adj(i, p)) for i, p in enumerate(ps)]) if ps else np.zeros(0, np.int64)
        self.base = cat("base", lambda i, p: np.where(p["base"] >= 0, sb[p["base_shard"]] + p["base"], -1)).astype(np.int64)
        self.kind = cat("kind").astype(np.uint8)
        self.stream_off = cat("stream_off", lambda i, p: p["stream_off"] + bb[i]).astype(np.int64)
        self.stream_len = cat("stream_len").astype(np.int64)
        self.raw_len = cat("raw_len").astype(np.int64)
        dep = dependency_order(self.base)       # records renumbered once so that a dictionary precedes its dependants
        if dep is not None:
            order, new_of_old = dep
            self.base = np.where(self.base >= 0, new_of_old[np.maximum(self.base, 0)], -1)[order]
            self.kind, self.stream_off, self.stream_len, self.raw_len = self.kind[order], self.stream_off[order], self.stream_len[order], self.raw_len[order]
        blobs = [t(m.blob, torch.uint8) for m in shards if m.blob.size]
        self.blob = blobs[0] if len(blobs) == 1 else torch.cat(blobs) if blobs else torch.zeros(1, dtype=torch.uint8, device=device)
        self.slot = np.concatenate([sb[m.chunk_map["shard"].astype(np.int64)] + m.chunk_map["slot"].astype(np.int64) for m in shards]) \
            if shards else np.zeros(0, np.int64)
        lens = np.concatenate([m.chunk_map["raw_length"].astype(np.int64) for m in shards]) if shards else np.zeros(0, np.int64)
        from .manifest import stream_order
        perm = stream_order(shards)             # a multi-rank stream's store: requests address the stream, whose chunks interleave the shards
        if perm is not None:
            self.slot, lens = self.slot[perm], lens[perm]
        self.cuts = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)          # chunk map: byte offset of every chunk
        self.sha = np.concatenate([m.index["sha256"] for m in shards]) if shards else np.zeros((0, 32), np.uint8)
        if dep is not None:
            self.slot, self.sha = new_of_old[self.slot], self.sha[order]
        self.n_bytes = int(self.cuts[-1])
        self.last = {}

    def closure(self, slots: np.ndarray) -> np.ndarray:
        """The stored slots needed to decode `slots`: themselves plus, transitively, the dictionaries of the DELTA records
        among them (ascending: a dictionary always precedes its dependants)."""
        need = np.unique(slots)
        frontier = need
        while frontier.size:
            b = self.base[frontier]
            b = np.unique(b[b >= 0])
            frontier = b[~np.isin(b, need, assume_unique=True)]
            if frontier.size:
                need = np.union1d(need, frontier)
        return need

    def read_ranges(self, ranges, verify: bool = True) -> list:
        """[(offset, length), ...] -> one uint8 tensor on the device per request (views into one buffer)."""
        dev = self.dev
        if not len(ranges):
            return []
        r = np.asarray(ranges, np.int64).reshape(-1, 2)
        if (r[:, 0] < 0).any() or (r[:, 1] < 0).any() or (r[:, 0] + r[:, 1] > self.n_bytes).any():
            raise ReadError("a requested range lies outside the stored corpus")
        lo = np.searchsorted(self.cuts, r[:, 0], side="right") - 1                     # first chunk of every request
        hi = np.searchsorted(self.cuts, r[:, 0] + r[:, 1], side="left")               # one past its last chunk
        hi = np.maximum(hi, lo)
        touched = np.zeros(len(self.cuts) - 1, bool)
        for a, b in zip(lo, hi):
            touched[a:b] = True
        chunks = np.nonzero(touched)[0]
        slots = self.slot[chunks]
        need = self.closure(slots)                                                   # ascending global slot ids
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
        b = self.base[need]
        base_sel = np.where(b >= 0, np.searchsorted(need, np.maximum(b, 0)), -1)      # dictionaries renumbered into the subset
        raw, raw_off, _ = ops.l1_inflate(self.blob, t(self.stream_off[need], torch.int64), t(self.kind[need], torch.uint8), t(base_sel, torch.int64),
                                         t(self.raw_len[need], torch.int64), stream_len=t(self.stream_len[need], torch.int32))
        # the touched chunks, back to back, from the slots that hold them (POINTER: the target's)
        clen = self.cuts[chunks + 1] - self.cuts[chunks]
        tcuts = np.concatenate([[0], np.cumsum(clen)]).astype(np.int64)
        tcuts_d = t(tcuts, torch.int64)
        buf = ops.read_assemble(tcuts_d, t(np.searchsorted(need, slots), torch.int64), raw_off, raw)
        if verify and len(self.sha) and self.sha.any():
            verify_digests(buf, tcuts_d, t(self.sha[slots], torch.uint8))
        pos = np.searchsorted(chunks, lo)                                              # request -> its first chunk's place in buf
        start = tcuts[np.minimum(pos, len(tcuts) - 1)] + (r[:, 0] - self.cuts[np.minimum(lo, len(self.cuts) - 2)])
        self.last = {"requests": len(r), "chunks_touched": int(len(chunks)), "records_decoded": int(len(need)),
                     "dictionaries_pulled_in": int(len(need) - len(np.unique(slots))), "bytes_decoded": int(self.raw_len[need].sum()),
                     "bytes_requested": int(r[:, 1].sum())}
        return [buf[int(s): int(s) + int(n)] for s, n in zip(start, r[:, 1])]


def read_ranges(store, ranges, device, verify: bool = True) -> list:
    """One-off form of StoreReader(store, device).read_ranges(ranges)."""
    return StoreReader(store, device).read_ranges(ranges, verify)
