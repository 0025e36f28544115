"""Chunk manifest: the reference's packed on-disk records, packed on the GPU from a ShardResult (hmse_manifest_pack).

Layouts (little-endian, packed) exactly as in the reference:
  ChunkIndex  40 B {sha256[32], lba u32, length u16, refcount u16}        README.md:1263-1270, 2646-2651
  DeltaChunk   8 B {base_lba u32, base_length u16, delta_length u16} + delta_data[]   README.md:2182-2189
  pointer      8 B {target_lba u32, target_length u16, flags u16}          README.md:1312 ("LBA + offset")
`lba` is a byte offset into the blob divided by `lba_unit` (the reference's LBA is a 512-byte SD
sector; the unit is stored in the header and every record in the blob is aligned to it).
The per-chunk map (README.md:1448 "Chunk Map [#301, #302d, ...]") is a u32 index-table slot, a type
tag and the shard that holds the slot.  A sharded run (one process per GPU) writes one Manifest per
shard; merge_manifests() resolves the pointers that cross shards into a Store.  `reconstruct()` is the three-branch read path of README.md:1621-1675, used
as the end-to-end verifier (VALIDATION_METHODS.md:257 "100 % lossless reconstruction required").
"""
from __future__ import annotations

import hashlib
import struct
import zlib
from dataclasses import dataclass

import numpy as np

from .config import KIND_DELTA, KIND_FULL, KIND_POINTER

MAGIC = b"HMSEMI35"
CHUNK_INDEX_DTYPE = np.dtype([("sha256", "u1", 32), ("lba", "<u4"), ("length", "<u2"), ("refcount", "<u2")])
DELTA_HDR_DTYPE = np.dtype([("base_lba", "<u4"), ("base_length", "<u2"), ("delta_length", "<u2")])
POINTER_DTYPE = np.dtype([("target_lba", "<u4"), ("target_length", "<u2"), ("flags", "<u2")])
REMOTE_BASE_DTYPE = np.dtype([("slot", "<u4"), ("shard", "<u4"), ("base_slot", "<u4")])   # DELTA record `slot` of this shard: dictionary = record `base_slot` of `shard`
MAP_DTYPE = np.dtype([("slot", "<u4"), ("raw_length", "<u2"), ("kind", "u1"), ("shard", "u1")])
PIECE_DTYPE = np.dtype([("g0", "<u8"), ("n", "<u8")])   # multi-rank stream: a run of `n` consecutive local chunks whose global (stream-order) indices start at g0
assert CHUNK_INDEX_DTYPE.itemsize == 40 and DELTA_HDR_DTYPE.itemsize == 8 and POINTER_DTYPE.itemsize == 8


@dataclass
class Manifest:
    """One shard's part of a store: its blob, its index, the map of ITS chunks (global chunk order = shard, local)."""
    lba_unit: int
    index: np.ndarray      # CHUNK_INDEX_DTYPE [n_unique]
    chunk_map: np.ndarray  # MAP_DTYPE [n_chunks]
    pointers: np.ndarray   # POINTER_DTYPE [n_pointer]
    blob: np.ndarray       # uint8: FULL streams and DeltaChunk records, lba_unit-aligned
    shard: int = 0
    n_shards: int = 1
    chunk_base: int = 0    # global index of this shard's chunk 0
    remote_bases: np.ndarray | None = None   # REMOTE_BASE_DTYPE: DELTA records whose dictionary is another shard's record (global L4)
    pieces: np.ndarray | None = None         # PIECE_DTYPE: this shard's chunks in STREAM order (multi-rank streams: the shards interleave)

    def n_remote(self) -> int:
        return 0 if self.remote_bases is None else len(self.remote_bases)

    def global_index(self) -> np.ndarray | None:
        """Stream-order index of every local chunk (None: the store's order is (shard, local))."""
        if self.pieces is None:
            return None
        return np.concatenate([np.arange(int(p["g0"]), int(p["g0"]) + int(p["n"]), dtype=np.int64) for p in self.pieces]) if len(self.pieces) else np.zeros(0, np.int64)

    def to_bytes(self) -> bytes:
        # version 2 = no record names a dictionary outside its own blob; version 3 appends the table of those that do
        # (the 8-byte DeltaChunk header of README.md:2182-2189 has no room for a shard number); version 4 (multi-rank stream)
        # appends, behind that table, the shard's pieces: its chunks' places in the stream order
        nr = self.n_remote()
        ver = 4 if self.pieces is not None else 3 if nr else 2
        hdr = MAGIC + struct.pack("<IIQQQQIIQ", ver, self.lba_unit, len(self.index), len(self.chunk_map), len(self.pointers), self.blob.size,
                                  self.shard, self.n_shards, self.chunk_base)
        tail = struct.pack("<Q", nr) + (self.remote_bases.tobytes() if nr else b"") if (nr or ver == 4) else b""
        if ver == 4:
            tail += struct.pack("<Q", len(self.pieces)) + self.pieces.tobytes()
        return hdr + self.index.tobytes() + self.chunk_map.tobytes() + self.pointers.tobytes() + self.blob.tobytes() + tail

    @staticmethod
    def from_bytes(b: bytes) -> "Manifest":
        assert b[:8] == MAGIC
        ver, unit, nu, nc, npt, nb, shard, n_shards, cbase = struct.unpack_from("<IIQQQQIIQ", b, 8)
        assert ver in (2, 3, 4)
        o = 8 + struct.calcsize("<IIQQQQIIQ")
        idx = np.frombuffer(b, CHUNK_INDEX_DTYPE, nu, o); o += nu * 40
        cmap = np.frombuffer(b, MAP_DTYPE, nc, o); o += nc * 8
        ptr = np.frombuffer(b, POINTER_DTYPE, npt, o); o += npt * 8
        blob = np.frombuffer(b, np.uint8, nb, o); o += nb
        remote = pieces = None
        if ver >= 3:
            (nr,) = struct.unpack_from("<Q", b, o)
            remote = np.frombuffer(b, REMOTE_BASE_DTYPE, nr, o + 8) if nr else None
            o += 8 + nr * REMOTE_BASE_DTYPE.itemsize
        if ver == 4:
            (npc,) = struct.unpack_from("<Q", b, o)
            pieces = np.frombuffer(b, PIECE_DTYPE, npc, o + 8)
        return Manifest(unit, idx, cmap, ptr, blob, shard, n_shards, cbase, remote, pieces)

    def nbytes(self) -> int:
        return 8 + struct.calcsize("<IIQQQQIIQ") + self.index.nbytes + self.chunk_map.nbytes + self.pointers.nbytes + self.blob.nbytes


PTR_UNRESOLVED = 0x8000   # pointer-record flag: the target lives on another shard and its lba is filled in by merge_manifests


def pack_manifest_device(res, shard: int = 0, n_shards: int = 1, any_target: bool = False):
    """The shard's packed records in HBM: (lba_unit, blob u8[], index u8[u,40], chunk_map u8[n,8], pointers u8[p,8]).
    ONE pass of GPU kernels over the ShardResult (hmse_manifest_pack, hmse_amd/csrc/manifest_pack.hip) plus two prefix
    sums for the record and pointer positions; the only host sync reads the two output sizes."""
    import torch

    from . import ops
    if not res.cuts.is_cuda:
        raise ops.HmseError(-1, "the manifest is packed on the GPU: the ShardResult must live in HBM")
    if res.streams is None:
        raise ops.HmseError(-1, "the manifest needs the L1 layer's streams")
    dev = res.cuts.device
    n = res.cuts.numel() - 1
    u = res.uniq_ids.numel()
    slen = res.stream_off[1:] - res.stream_off[:-1]
    rec_len = slen + 8 * (res.kind == KIND_DELTA).to(torch.int64)
    mine = torch.arange(res.chunk_base, res.chunk_base + n, dtype=torch.int64, device=dev)
    is_ptr = (res.first_occ != mine) if res.first_occ is not None else torch.zeros(n, dtype=torch.bool, device=dev)
    ptr_index = torch.cumsum(is_ptr.to(torch.int64), 0) - is_ptr.to(torch.int64)
    total, n_ptr = (int(v) for v in torch.stack([rec_len.sum(), is_ptr.sum()]).tolist())   # the one host sync: sizes of the outputs
    unit = 1
    while (total + unit * u) // unit >= 2**32:   # lba is 32 bits (README.md:1266): coarser units for blobs beyond 4 GiB
        unit *= 2
    rec_off = torch.zeros(u + 1, dtype=torch.int64, device=dev)
    torch.cumsum((rec_len + (unit - 1)) // unit * unit, 0, out=rec_off[1:])
    blob_bytes = total if unit == 1 else int(rec_off[-1].item())
    blob = torch.empty(blob_bytes, dtype=torch.uint8, device=dev)
    index = torch.empty((u, 40), dtype=torch.uint8, device=dev)
    cmap = torch.empty((n, 8), dtype=torch.uint8, device=dev)
    ptrs = torch.empty((n_ptr, 8), dtype=torch.uint8, device=dev)
    ops.manifest_pack(res, shard, n_shards, getattr(res, "shard_bases", None), rec_off, unit, ptr_index, blob, index, cmap, ptrs, any_target=any_target)
    return unit, blob, index, cmap, ptrs


def remote_base_table(res) -> np.ndarray | None:
    """DELTA records of a global-L4 shard whose dictionary is stored on another shard: (slot, shard, that shard's slot).
    Their DeltaChunk headers are packed unresolved (base_lba 0xFFFFFFFF); merge_manifests() fills them in."""
    import torch
    if hasattr(res, "remote_bases"):      # a global-L4 STREAM's shard: its stored chunks interleave with the other ranks' in the
        return res.remote_bases           # global numbering, the owner table comes with the result (stream_dist.GlobalL4StreamIngest)
    bg = getattr(res, "base_global", None)
    if bg is None:
        return None
    remote = (bg >= 0) & (res.base < 0) & (res.kind == KIND_DELTA)
    slots = remote.nonzero().flatten()
    if slots.numel() == 0:
        return None
    ub = torch.as_tensor(list(res.u_bases), dtype=torch.int64, device=bg.device)
    g = bg[slots]
    sh = torch.searchsorted(ub, g, right=True) - 1
    out = np.zeros(slots.numel(), REMOTE_BASE_DTYPE)
    out["slot"] = slots.cpu().numpy(); out["shard"] = sh.cpu().numpy(); out["base_slot"] = (g - ub[sh]).cpu().numpy()
    return out


def build_manifest(res, shard: int = 0, n_shards: int = 1) -> Manifest:
    """pack_manifest_device() + one device -> host copy per array: the host receives four finished arrays and only has to
    write() them.  A chunk whose first occurrence lives on another shard (sharded ingest) becomes a POINTER
    with an unresolved pointer record; merge_manifests() resolves those once every shard's manifest exists."""
    pieces = getattr(res, "pieces", None)           # a multi-rank stream's shard (stream_dist.store_results): stream-order table,
    unit, blob, index, cmap, ptrs = pack_manifest_device(res, shard, n_shards, any_target=pieces is not None)   # targets on any shard
    host = lambda t, dt: np.frombuffer(t.cpu().numpy().tobytes(), dt)
    return Manifest(unit, host(index, CHUNK_INDEX_DTYPE), host(cmap, MAP_DTYPE), host(ptrs, POINTER_DTYPE), blob.cpu().numpy(),
                    shard, n_shards, int(res.chunk_base), remote_base_table(res), pieces)


@dataclass
class Store:
    """A sharded store: one Manifest per shard, cross-shard pointers resolved (global chunk order = shard, local index)."""
    shards: list

    def to_bytes(self) -> bytes:
        parts = [m.to_bytes() for m in self.shards]
        return b"HMSESTOR" + struct.pack("<I", len(parts)) + b"".join(struct.pack("<Q", len(p)) + p for p in parts)

    @staticmethod
    def from_bytes(b: bytes) -> "Store":
        assert b[:8] == b"HMSESTOR"
        (k,) = struct.unpack_from("<I", b, 8)
        o, shards = 12, []
        for _ in range(k):
            (ln,) = struct.unpack_from("<Q", b, o)
            shards.append(Manifest.from_bytes(b[o + 8:o + 8 + ln])); o += 8 + ln
        return Store(shards)


def merge_manifests(parts: list) -> Store:
    """The global index of a sharded run: per-shard blobs and indexes stay as written; every POINTER whose target lives on
    another shard gets its map slot (that shard's index slot) and its pointer record {target_lba, target_length,
    flags = POINTER | shard << 4} from the target shard's own records (README.md:1312 "LBA + offset", 1635-1669)."""
    parts = sorted(parts, key=lambda m: m.shard)
    assert [m.shard for m in parts] == list(range(len(parts))) and all(m.n_shards == len(parts) for m in parts), "one manifest per shard"
    out = []
    for m in parts:
        cmap, ptrs = m.chunk_map.copy(), m.pointers.copy()
        is_ptr = cmap["kind"] == KIND_POINTER
        unresolved = (ptrs["flags"] & PTR_UNRESOLVED) != 0
        pidx = np.nonzero(is_ptr)[0][unresolved]                     # chunk index of every unresolved pointer
        for r in np.unique(cmap["shard"][pidx]):
            t = parts[int(r)]
            assert int(r) < m.shard or (m.pieces is not None and int(r) != m.shard), "dedupe only ever points backwards (the shards of a stream interleave: any other shard)"
            sel = pidx[cmap["shard"][pidx] == r]
            tgt_chunk = cmap["slot"][sel].astype(np.int64)           # the target shard's LOCAL chunk index
            assert (tgt_chunk < len(t.chunk_map)).all() and (t.chunk_map["kind"][tgt_chunk] != KIND_POINTER).all(), \
                "a cross-shard pointer must name a stored chunk of its target shard"
            slot = t.chunk_map["slot"][tgt_chunk]
            cmap["slot"][sel] = slot
            rec = np.nonzero(unresolved)[0][cmap["shard"][pidx] == r]
            ptrs["target_lba"][rec] = t.index["lba"][slot]
            ptrs["target_length"][rec] = t.index["length"][slot]
            ptrs["flags"][rec] = KIND_POINTER | (int(r) << 4)
        blob = m.blob
        if m.n_remote():
            # DeltaChunk headers whose dictionary lives in another shard's blob (an earlier one; any other one in a stream's store): {base_lba, base_length} from that shard's index
            # (which shard: the manifest's remote_bases table — the 8-byte header has no room for it)
            blob = m.blob.copy()
            rb = m.remote_bases
            assert ((rb["shard"] < m.shard) | ((rb["shard"] != m.shard) & (m.pieces is not None))).all(), \
                "a dictionary is always an EARLIER stored chunk (the shards of a stream interleave: any other shard)"
            for r in np.unique(rb["shard"]):
                t = parts[int(r)]
                sel = rb[rb["shard"] == r]
                pos = m.index["lba"][sel["slot"]].astype(np.int64) * m.lba_unit
                h = np.zeros(len(sel), DELTA_HDR_DTYPE)
                h["base_lba"] = t.index["lba"][sel["base_slot"]]; h["base_length"] = t.index["length"][sel["base_slot"]]
                h["delta_length"] = m.index["length"][sel["slot"]] - 8
                blob[pos[:, None] + np.arange(8)[None, :]] = np.frombuffer(h.tobytes(), np.uint8).reshape(len(sel), 8)
        out.append(Manifest(m.lba_unit, m.index, cmap, ptrs, blob, m.shard, m.n_shards, m.chunk_base, m.remote_bases, m.pieces))
    return Store(out)


def stream_order(shards: list) -> np.ndarray | None:
    """A store written by a multi-rank STREAM keeps its chunks per shard, but the original bytes are the chunks in stream
    order (batch, rank, local): the permutation from stream position to the (shard, local)-concatenated chunk list, or None for
    a store whose order is (shard, local)."""
    if not shards or all(m.pieces is None for m in shards):
        return None
    assert all(m.pieces is not None for m in shards), "every shard of a stream store carries its pieces"
    g = np.concatenate([m.global_index() for m in shards])
    assert len(g) == sum(len(m.chunk_map) for m in shards)
    perm = np.argsort(g, kind="stable")
    assert np.array_equal(g[perm], np.arange(len(g))), "the shards' pieces must tile the stream"
    return perm


def reconstruct(m) -> bytes:
    """Read path (README.md:1621-1675) in plain Python + stock zlib — the end-to-end verifier: FULL -> inflate;
    POINTER -> target (possibly on another shard); DELTA -> inflate with zdict = base.  Takes a Manifest or a Store."""
    shards = m.shards if isinstance(m, Store) else [m]
    if any(((s.pointers["flags"] & PTR_UNRESOLVED) != 0).any() for s in shards):
        raise ValueError("the store has unresolved cross-shard pointers: merge_manifests() its shards first")
    by_lba = [{int(e["lba"]): i for i, e in enumerate(s.index)} for s in shards]
    slot_kind = []
    for s in shards:
        k = np.zeros(len(s.index), np.uint8)
        own = s.chunk_map["kind"] != KIND_POINTER
        k[s.chunk_map["slot"][own]] = s.chunk_map["kind"][own]
        slot_kind.append(k)
    remote = [{int(e["slot"]): (int(e["shard"]), int(e["base_slot"])) for e in (s.remote_bases if s.remote_bases is not None else [])} for s in shards]
    cache: dict[tuple[int, int], bytes] = {}

    def raw_of(r: int, slot: int) -> bytes:
        if (r, slot) in cache:
            return cache[(r, slot)]
        s = shards[r]
        e = s.index[slot]
        o = int(e["lba"]) * s.lba_unit
        rec = s.blob[o:o + int(e["length"])].tobytes()
        if slot_kind[r][slot] == KIND_DELTA:  # DeltaChunk record: 8-byte header, then a stream with the base as dictionary
            base_lba, base_len, dlen = struct.unpack_from("<IHH", rec, 0)
            br, bslot = remote[r][slot] if slot in remote[r] else (r, by_lba[r][base_lba])   # the dictionary may be another shard's record
            if br != r and (base_lba == 0xFFFFFFFF or int(shards[br].index["lba"][bslot]) != base_lba):
                raise ValueError("a cross-shard DeltaChunk header is unresolved: merge_manifests() the shards first")
            d = zlib.decompressobj(-15, zdict=raw_of(br, bslot))
            out = d.decompress(rec[8:8 + dlen]) + d.flush()
        else:
            d = zlib.decompressobj(-15)
            out = d.decompress(rec) + d.flush()
        assert d.eof and not d.unused_data
        if e["sha256"].any():
            assert hashlib.sha256(out).digest() == e["sha256"].tobytes(), "SHA-256 mismatch on reconstruct"
        cache[(r, slot)] = out
        return out

    chunks = [(int(c["shard"]), int(c["slot"])) for s in shards for c in s.chunk_map]
    perm = stream_order(shards)
    if perm is not None:
        chunks = [chunks[int(i)] for i in perm]
    return b"".join(raw_of(r, slot) for r, slot in chunks)
