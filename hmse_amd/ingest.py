"""The L1-L4 ingest pipeline on one GPU shard (one process per GPU) — host driver over ops.py.

Execution order per SURVEY.md D1 / BASELINE.json: L2 FastCDC -> L3 SHA-256 (+ RCCL all-gather of
digests, global first-occurrence dedupe) -> L4 MinHash/LSH on unique chunks -> L1 dictionary DEFLATE.
Stage boundaries and record formats stay those of the reference (README.md:286-291, 1263-1270,
2182-2189).  The corpus shards by whole segments; the only collective is the digest all-gather
(SURVEY.md §8e): every rank then evaluates the same deterministic first-occurrence rule on the same
gathered array, so dedupe is bit-identical to the 1-GPU run.  L4 base selection is scoped to the
local shard by default (the dictionary bytes must be resident); `global_l4` adds a signature all-gather
and a cross-GPU fetch of remote base chunks and makes the whole result that of the 1-GPU run.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import torch

from . import ops
from .config import LAYER_L1, LAYER_L2, LAYER_L3, LAYER_L4, IngestConfig


@dataclass
class ShardResult:
    n_bytes: int
    cuts: torch.Tensor                 # int64 [n+1] local chunk ends
    digests: torch.Tensor | None       # uint8 [n, 32]
    chunk_base: int                    # global index of local chunk 0
    n_global: int                      # chunks on all ranks
    first_occ: torch.Tensor | None     # int64 [n] global index of the earliest equal chunk
    refcount: torch.Tensor | None      # int32 [n] (non-zero at global first occurrences)
    uniq_ids: torch.Tensor             # int64 [u] local chunk ids that are stored (first occurrences)
    sig: torch.Tensor | None           # int32 [u, 128]
    band_keys: torch.Tensor | None     # int32 [u, bands]
    base: torch.Tensor | None          # int64 [u] index into uniq_ids of the dictionary chunk, -1 none
    streams: torch.Tensor | None       # uint8 dense DEFLATE streams of the stored chunks
    stream_off: torch.Tensor | None    # int64 [u+1]
    kind: torch.Tensor | None          # uint8 [u] FULL / DELTA
    stats: dict = field(default_factory=dict)
    shard_bases: list | None = None    # sharded runs: chunk_base of every shard (what the manifest needs to name a target shard)
    base_global: torch.Tensor | None = None   # global L4: int64 [u] GLOBAL stored-chunk index of the dictionary, -1 none
    u_base: int = 0                    # global L4: global stored-chunk index of this shard's first stored chunk
    u_bases: list | None = None        # global L4: u_base of every shard


def fixed_cuts(n: int, cfg: IngestConfig, seg_off: torch.Tensor) -> torch.Tensor:
    """L2 disabled (ablation 'L1 only', VALIDATION_METHODS.md:458): fixed max_size blocks per segment."""
    so = seg_off.tolist()
    pieces = [torch.arange(a, b, cfg.max_size, dtype=torch.int64, device=seg_off.device) for a, b in zip(so[:-1], so[1:]) if b > a]
    cuts = torch.cat(pieces + [torch.tensor([n], dtype=torch.int64, device=seg_off.device)]) if pieces else \
        torch.zeros(1, dtype=torch.int64, device=seg_off.device)
    return cuts


def gather_rows(rows: torch.Tensor, group=None):
    """All-gather of a per-rank table [n_r, ...] (counts first, then the rows padded to the longest rank) over RCCL/xGMI.

    Returns (all rows in (rank, local) order, first global row of this rank, total rows, first global row of every rank)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = rows.device
    # the collective runs where the backend moves bytes: on the device for RCCL, through host memory for gloo (the CPU
    # tests, and a rehearsal of an N-rank run with every rank on one GPU: bench.py HMSE_BENCH_REHEARSE)
    xdev = torch.device("cpu") if dist.get_backend(group) == "gloo" else dev
    n_loc = torch.tensor([rows.shape[0]], dtype=torch.int64, device=xdev)
    counts = torch.empty(world, dtype=torch.int64, device=xdev)
    dist.all_gather_into_tensor(counts, n_loc, group=group)
    cl = counts.tolist()
    mx = max(cl)
    if mx == 0:      # nobody has a row (every rank knows it from the counts): no second collective
        return rows.to(dev)[:0], 0, 0, [0] * world
    padded = torch.zeros((mx,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=xdev)
    padded[: rows.shape[0]] = rows.to(xdev)
    allp = torch.empty((world * mx,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=xdev)
    dist.all_gather_into_tensor(allp, padded, group=group)
    allp = allp.to(dev)
    if all(c == mx for c in cl):
        allr = allp
    else:
        allr = torch.cat([allp[r * mx: r * mx + cl[r]] for r in range(world)])
    bases = [sum(cl[:r]) for r in range(world)]
    return allr, bases[rank], sum(cl), bases


def gather_digests(digests: torch.Tensor, group=None):
    """The one data-path collective of the default pipeline: all-gather of (count, digests) over RCCL/xGMI.

    Returns (all_digests [N, 32] in (rank, local) order, chunk_base of this rank, N, chunk_base of every rank)."""
    return gather_rows(digests, group)


def route_requests(req: torch.Tensor, u_bases: list, world: int):
    """Global stored-chunk ids (ascending) -> per-owner request counts and the owners' local indices, in request order."""
    ub = torch.tensor(list(u_bases), dtype=torch.int64, device=req.device)
    owner = torch.searchsorted(ub, req, right=True) - 1
    counts = torch.bincount(owner, minlength=world)[:world]
    return counts, req - ub[owner]


def fetch_chunks(req: torch.Tensor, u_bases: list, data: torch.Tensor, cuts: torch.Tensor, uniq_ids: torch.Tensor, group=None,
                 gather=None):
    """Cross-GPU base fetch (SURVEY.md §8e last sentence, §8f-3): raw bytes of the stored chunks `req` (global stored-chunk
    ids, ascending) from the ranks that hold them.  Three all-to-alls over RCCL/xGMI: request counts, requested ids,
    then lengths + bytes (each owner gathers the requested chunks into one send buffer with hmse_read_assemble).
    Every rank calls it (with an empty `req` if it needs nothing).  Returns (bytes uint8[sum], lens int64[len(req)]).
    `gather(out_cuts, chunk_ids, cuts, data)` stands in for the HIP gather in the gloo test of the exchange itself."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    counts, local_idx = route_requests(req, u_bases, world)
    return fetch_chunks_routed(counts, local_idx, data, cuts, uniq_ids, group, gather)


def fetch_chunks_routed(counts: torch.Tensor, local_idx: torch.Tensor, data: torch.Tensor, cuts: torch.Tensor, uniq_ids: torch.Tensor,
                        group=None, gather=None):
    """fetch_chunks() with the routing done by the caller: `counts[r]` requests go to rank r, `local_idx` holds the owners' local
    stored-chunk indices grouped by owner in rank order (a multi-rank STREAM's stored chunks interleave in the global numbering, so
    the owner is looked up, not computed from a base table: stream_dist.GlobalL4StreamIngest).  Returns (bytes, lens) in that order."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = data.device
    xdev = torch.device("cpu") if dist.get_backend(group) == "gloo" else dev
    want = counts.to(xdev)
    asked = torch.empty(world, dtype=torch.int64, device=xdev)
    dist.all_to_all_single(asked, want, group=group)                    # how many chunks each rank asks of me
    wl, al = want.tolist(), asked.tolist()
    ids_in = torch.empty(sum(al), dtype=torch.int64, device=xdev)
    dist.all_to_all_single(ids_in, local_idx.to(xdev), output_split_sizes=al, input_split_sizes=wl, group=group)
    # serve: the requested chunks' bytes, in request order (ids come from other ranks: checked before they index anything)
    if ids_in.numel() and (int(ids_in.min()) < 0 or int(ids_in.max()) >= uniq_ids.numel()):
        raise ValueError(f"fetch_chunks: a peer asked for stored chunk {int(ids_in.max())} of {uniq_ids.numel()} (mismatched u_bases?)")
    cid = uniq_ids[ids_in.to(dev)]
    lens_out = cuts[cid + 1] - cuts[cid]
    out_cuts = torch.zeros(cid.numel() + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lens_out, 0, out=out_cuts[1:])
    payload = (gather or ops.read_assemble)(out_cuts, cid, cuts, data) if cid.numel() else torch.empty(0, dtype=torch.uint8, device=dev)
    lens_in = torch.empty(sum(wl), dtype=torch.int64, device=xdev)
    dist.all_to_all_single(lens_in, lens_out.to(xdev), output_split_sizes=wl, input_split_sizes=al, group=group)
    oc = out_cuts.tolist()
    send_b = [oc[sum(al[:r + 1])] - oc[sum(al[:r])] for r in range(world)]
    li = lens_in.tolist()
    recv_b = [sum(li[sum(wl[:r]): sum(wl[:r + 1])]) for r in range(world)]
    if (lens_in < 0).any() or (lens_in > 65536).any():
        raise ValueError("fetch_chunks: a peer announced an impossible chunk length")
    got = torch.empty(sum(recv_b), dtype=torch.uint8, device=xdev)
    dist.all_to_all_single(got, payload.to(xdev), output_split_sizes=recv_b, input_split_sizes=send_b, group=group)
    return got.to(dev), lens_in.to(dev)


def ingest_shard(data: torch.Tensor, cfg: IngestConfig, seg_off: torch.Tensor | None = None, group=None,
                 distributed: bool = False, want_stats: bool = True, exchange=None, pre=None,
                 global_l4: bool = False, sig_exchange=None, chunk_fetch=None) -> ShardResult:
    """Run the enabled layers over this rank's shard (device-resident uint8 tensor).

    `exchange(digests) -> (all_digests, chunk_base, n_global, shard_bases)` replaces the RCCL all-gather (used to run the
    shards of a sharded store one after another on a single GPU); `pre = (cuts, digests[, sig])` skips L2/L3 (and L4a) when
    the caller has already run them for that exchange.

    `global_l4` (sharded runs): base selection over ALL shards instead of the local one — the signatures are all-gathered
    like the digests (`sig_exchange(sig) -> (sig_all, u_base, U, u_bases)` replaces that collective), every rank runs the
    same LSH over the same array, and a base chunk stored on another rank is fetched over xGMI (`fetch_chunks`, or
    `chunk_fetch(req) -> (bytes, lens)`) and appended behind the shard as a ghost chunk that the DEFLATE kernel addresses by
    chunk id.  Dedupe and base selection are then those of the 1-GPU run: the N-shard streams are bit-identical to its
    streams, the CF drift of shard-local L4 is gone.  `res.base_global` names each record's dictionary as a global
    stored-chunk index; records with a remote dictionary decode with `read.reconstruct_shards` (all shards together)."""
    n = data.numel()
    dev = data.device
    if seg_off is None:
        seg_off = ops.segment_offsets(n, cfg.seg_size, dev)
    # L2
    if pre is not None:
        cuts = pre[0]
    else:
        cuts = ops.l2_cdc(data, cfg, seg_off) if cfg.layers & LAYER_L2 else fixed_cuts(n, cfg, seg_off)
    n_chunks = cuts.numel() - 1
    digests = first_occ = refcount = None
    chunk_base, n_global, shard_bases = 0, n_chunks, None
    # L3
    if cfg.layers & LAYER_L3:
        digests = pre[1] if pre is not None else ops.l3_sha256(data, cuts)
        if distributed or exchange is not None:
            alld, chunk_base, n_global, shard_bases = (exchange or (lambda d: gather_digests(d, group)))(digests)
            fo_all, rc_all = ops.l3_dedup(alld)
            first_occ = fo_all[chunk_base: chunk_base + n_chunks]
            refcount = rc_all[chunk_base: chunk_base + n_chunks]
        else:
            first_occ, refcount = ops.l3_dedup(digests)
        mine = torch.arange(chunk_base, chunk_base + n_chunks, dtype=torch.int64, device=dev)
        uniq_ids = (first_occ == mine).nonzero().flatten()
    else:
        uniq_ids = torch.arange(n_chunks, dtype=torch.int64, device=dev)
    # L4
    sig = band_keys = base = base_global = None
    base_chunk = None           # global L4: dictionary as a chunk id into (cuts ++ ghost chunks)
    data_x, cuts_x = data, cuts
    u_base, u_bases = 0, None
    if cfg.layers & LAYER_L4:
        sig = pre[2] if pre is not None and len(pre) > 2 else ops.l4_minhash(data, cuts, cfg, uniq_ids)
        if global_l4 and (distributed or sig_exchange is not None):
            u = uniq_ids.numel()
            sig_all, u_base, _, u_bases = (sig_exchange or (lambda sg: gather_rows(sg, group)))(sig)
            keys_all, base_all = ops.l4_lsh(sig_all, cfg)
            band_keys = keys_all[u_base: u_base + u]
            base_global = base_all[u_base: u_base + u]
            local = (base_global >= u_base) & (base_global < u_base + u)
            remote = (base_global >= 0) & ~local
            base = torch.where(local, base_global - u_base, torch.full_like(base_global, -1))
            base_chunk = torch.where(local, uniq_ids[base.clamp(min=0)], torch.full_like(base_global, -1))
            req = torch.unique(base_global[remote])                      # ascending global stored-chunk ids
            ghost, glens = (chunk_fetch or (lambda rq: fetch_chunks(rq, u_bases, data, cuts, uniq_ids, group)))(req)
            if req.numel():
                data_x = torch.cat([data, ghost])
                cuts_x = torch.cat([cuts, n + torch.cumsum(glens, 0)])
                base_chunk = torch.where(remote, n_chunks + torch.searchsorted(req, base_global.clamp(min=0)), base_chunk)
        else:
            band_keys, base = ops.l4_lsh(sig, cfg)
    # L1
    streams = stream_off = kind = None
    if cfg.layers & LAYER_L1:
        if base_chunk is not None:
            streams, stream_off, kind = ops.l1_deflate(data_x, cuts_x, cfg, uniq_ids, base_chunk, base_is_chunk_id=True)
        else:
            streams, stream_off, kind = ops.l1_deflate(data, cuts, cfg, uniq_ids, base)
    del data_x
    res = ShardResult(n, cuts, digests, chunk_base, n_global, first_occ, refcount, uniq_ids, sig, band_keys, base,
                      streams, stream_off, kind, shard_bases=shard_bases)
    res.base_global, res.u_base, res.u_bases = base_global, u_base, u_bases
    if want_stats:
        res.stats = shard_stats(res)
    return res


def ingest_shards_local(shards: list, cfg: IngestConfig, global_l4: bool = False, seg_offs: list | None = None) -> list:
    """A sharded ingest with every shard on THIS GPU, one after the other: L2/L3 of every shard first, then the digest
    exchange as the all-gather would deliver it (concatenation in shard order), then L4/L1 per shard.  Each result is what
    the rank that owns the shard would hold after ingest_shard(distributed=True) — used to build and verify a multi-shard
    store without a second GPU, and by a single process that drives several shards."""
    # (`seg_offs`: per shard its own segment offsets, e.g. document-aligned ones from hmse_amd.partition, instead of fixed seg_size runs)
    pre = []
    for i, d in enumerate(shards):
        so = seg_offs[i] if seg_offs is not None else None
        cuts = ops.l2_cdc(d, cfg, so) if cfg.layers & LAYER_L2 else \
            fixed_cuts(d.numel(), cfg, so if so is not None else ops.segment_offsets(d.numel(), cfg.seg_size, d.device))
        pre.append((cuts, ops.l3_sha256(d, cuts) if cfg.layers & LAYER_L3 else None))
    if not cfg.layers & LAYER_L3:
        return [ingest_shard(d, cfg, pre=p) for d, p in zip(shards, pre)]
    counts = [p[0].numel() - 1 for p in pre]
    bases = [sum(counts[:r]) for r in range(len(shards))]
    alld = torch.cat([p[1] for p in pre])
    ex = [lambda _dg, r=r: (alld, bases[r], sum(counts), bases) for r in range(len(shards))]
    if not (global_l4 and cfg.layers & LAYER_L4):
        return [ingest_shard(d, cfg, pre=p, exchange=ex[r]) for r, (d, p) in enumerate(zip(shards, pre))]
    # global L4: the signatures of every shard's stored chunks first (what the signature all-gather would deliver), then
    # per shard the global base selection, with the ghost chunks taken straight from the owning shard's tensor
    fo_all, _ = ops.l3_dedup(alld)
    uniqs, sigs = [], []
    for r, (d, p) in enumerate(zip(shards, pre)):
        mine = torch.arange(bases[r], bases[r] + counts[r], dtype=torch.int64, device=d.device)
        uq = (fo_all[bases[r]: bases[r] + counts[r]] == mine).nonzero().flatten()
        uniqs.append(uq)
        sigs.append(ops.l4_minhash(d, p[0], cfg, uq))
    ucounts = [int(u.numel()) for u in uniqs]
    u_bases = [sum(ucounts[:r]) for r in range(len(shards))]
    sig_all = torch.cat(sigs)

    def fetch(req):
        counts_r, local_idx = route_requests(req, u_bases, len(shards))
        pieces, lens, o = [], [], 0
        for r, c in enumerate(counts_r.tolist()):
            if not c:
                continue
            cid = uniqs[r][local_idx[o: o + c]]
            ln = pre[r][0][cid + 1] - pre[r][0][cid]
            oc = torch.zeros(c + 1, dtype=torch.int64, device=req.device)
            torch.cumsum(ln, 0, out=oc[1:])
            pieces.append(ops.read_assemble(oc, cid, pre[r][0], shards[r])); lens.append(ln); o += c
        if not pieces:
            return torch.empty(0, dtype=torch.uint8, device=req.device), torch.empty(0, dtype=torch.int64, device=req.device)
        return torch.cat(pieces), torch.cat(lens)

    return [ingest_shard(d, cfg, pre=(p[0], p[1], sigs[r]), exchange=ex[r], global_l4=True,
                         sig_exchange=lambda _sg, r=r: (sig_all, u_bases[r], sum(ucounts), u_bases), chunk_fetch=fetch)
            for r, (d, p) in enumerate(zip(shards, pre))]


def shard_stats(r: ShardResult) -> dict:
    """Counts for the CF of SURVEY.md §8d: N_in / (stored + 40*unique + 8*pointer + 8*delta)."""
    n_chunks = r.cuts.numel() - 1
    n_unique = int(r.uniq_ids.numel())
    n_delta = int((r.kind == 2).sum().item()) if r.kind is not None else 0
    lens = r.cuts[1:] - r.cuts[:-1]
    unique_bytes = int(lens[r.uniq_ids].sum().item()) if n_unique else 0
    stored = int(r.streams.numel()) if r.streams is not None else unique_bytes
    bsel = r.base_global if getattr(r, "base_global", None) is not None else r.base
    lsh_hits = int((bsel >= 0).sum().item()) if bsel is not None else 0
    return {"bytes": r.n_bytes, "chunks": n_chunks, "unique": n_unique, "pointer": n_chunks - n_unique, "delta": n_delta,
            "lsh_hits": lsh_hits, "unique_bytes": unique_bytes, "stored_bytes": stored}


def merge_stats(stats: list[dict]) -> dict:
    tot = {k: sum(s[k] for s in stats) for k in stats[0]}
    overhead = 40 * tot["unique"] + 8 * tot["pointer"] + 8 * tot["delta"]
    tot["cf"] = tot["bytes"] / max(1, tot["stored_bytes"] + overhead)
    tot["cf_payload"] = tot["bytes"] / max(1, tot["stored_bytes"])
    tot["unique_chunk_ratio"] = tot["unique"] / max(1, tot["chunks"])
    tot["lsh_hit_rate"] = tot["lsh_hits"] / max(1, tot["unique"])
    tot["delta_rate"] = tot["delta"] / max(1, tot["unique"])
    return tot
