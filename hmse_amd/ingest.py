"""The L1-L4 ingest pipeline on one GPU shard (one process per GPU) — host driver over ops.py.

Execution order per SURVEY.md D1 / BASELINE.json: L2 FastCDC -> L3 SHA-256 (+ RCCL all-gather of
digests, global first-occurrence dedupe) -> L4 MinHash/LSH on unique chunks -> L1 dictionary DEFLATE.
Stage boundaries and record formats stay those of the reference (README.md:286-291, 1263-1270,
2182-2189).  The corpus shards by whole segments; the only collective is the digest all-gather
(SURVEY.md §8e): every rank then evaluates the same deterministic first-occurrence rule on the same
gathered array, so dedupe is bit-identical to the 1-GPU run.  L4 base selection is scoped to the
local shard (the dictionary bytes must be resident).
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


def fixed_cuts(n: int, cfg: IngestConfig, seg_off: torch.Tensor) -> torch.Tensor:
    """L2 disabled (ablation 'L1 only', VALIDATION_METHODS.md:458): fixed max_size blocks per segment."""
    so = seg_off.tolist()
    pieces = [torch.arange(a, b, cfg.max_size, dtype=torch.int64, device=seg_off.device) for a, b in zip(so[:-1], so[1:]) if b > a]
    cuts = torch.cat(pieces + [torch.tensor([n], dtype=torch.int64, device=seg_off.device)]) if pieces else \
        torch.zeros(1, dtype=torch.int64, device=seg_off.device)
    return cuts


def gather_digests(digests: torch.Tensor, group=None):
    """The one data-path collective: all-gather of (count, digests) over RCCL/xGMI.

    Returns (all_digests [N, 32] in (rank, local) order, chunk_base of this rank, N, chunk_base of every rank)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = digests.device
    # the collective runs where the backend moves bytes: on the device for RCCL, through host memory for gloo (the CPU
    # tests, and a rehearsal of an N-rank run with every rank on one GPU: bench.py HMSE_BENCH_REHEARSE)
    xdev = torch.device("cpu") if dist.get_backend(group) == "gloo" else dev
    n_loc = torch.tensor([digests.shape[0]], dtype=torch.int64, device=xdev)
    counts = torch.empty(world, dtype=torch.int64, device=xdev)
    dist.all_gather_into_tensor(counts, n_loc, group=group)
    cl = counts.tolist()
    mx = max(cl)
    padded = torch.zeros((mx, 32), dtype=torch.uint8, device=xdev)
    padded[: digests.shape[0]] = digests.to(xdev)
    allp = torch.empty((world * mx, 32), dtype=torch.uint8, device=xdev)
    dist.all_gather_into_tensor(allp, padded, group=group)
    allp = allp.to(dev)
    if all(c == mx for c in cl):
        alld = allp
    else:
        alld = torch.cat([allp[r * mx: r * mx + cl[r]] for r in range(world)])
    bases = [sum(cl[:r]) for r in range(world)]
    return alld, bases[rank], sum(cl), bases


def ingest_shard(data: torch.Tensor, cfg: IngestConfig, seg_off: torch.Tensor | None = None, group=None,
                 distributed: bool = False, want_stats: bool = True, exchange=None, pre=None) -> ShardResult:
    """Run the enabled layers over this rank's shard (device-resident uint8 tensor).

    `exchange(digests) -> (all_digests, chunk_base, n_global, shard_bases)` replaces the RCCL all-gather (used to run the
    shards of a sharded store one after another on a single GPU); `pre = (cuts, digests)` skips L2/L3 when the caller has
    already run them for that exchange."""
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
    sig = band_keys = base = None
    if cfg.layers & LAYER_L4:
        sig = ops.l4_minhash(data, cuts, cfg, uniq_ids)
        band_keys, base = ops.l4_lsh(sig, cfg)
    # L1
    streams = stream_off = kind = None
    if cfg.layers & LAYER_L1:
        streams, stream_off, kind = ops.l1_deflate(data, cuts, cfg, uniq_ids, base)
    res = ShardResult(n, cuts, digests, chunk_base, n_global, first_occ, refcount, uniq_ids, sig, band_keys, base,
                      streams, stream_off, kind, shard_bases=shard_bases)
    if want_stats:
        res.stats = shard_stats(res)
    return res


def ingest_shards_local(shards: list, cfg: IngestConfig) -> list:
    """A sharded ingest with every shard on THIS GPU, one after the other: L2/L3 of every shard first, then the digest
    exchange as the all-gather would deliver it (concatenation in shard order), then L4/L1 per shard.  Each result is what
    the rank that owns the shard would hold after ingest_shard(distributed=True) — used to build and verify a multi-shard
    store without a second GPU, and by a single process that drives several shards."""
    pre = []
    for d in shards:
        cuts = ops.l2_cdc(d, cfg) if cfg.layers & LAYER_L2 else fixed_cuts(d.numel(), cfg, ops.segment_offsets(d.numel(), cfg.seg_size, d.device))
        pre.append((cuts, ops.l3_sha256(d, cuts) if cfg.layers & LAYER_L3 else None))
    if not cfg.layers & LAYER_L3:
        return [ingest_shard(d, cfg, pre=p) for d, p in zip(shards, pre)]
    counts = [p[0].numel() - 1 for p in pre]
    bases = [sum(counts[:r]) for r in range(len(shards))]
    alld = torch.cat([p[1] for p in pre])
    return [ingest_shard(d, cfg, pre=p, exchange=lambda _dg, r=r: (alld, bases[r], sum(counts), bases)) for r, (d, p) in enumerate(zip(shards, pre))]


def shard_stats(r: ShardResult) -> dict:
    """Counts for the CF of SURVEY.md §8d: N_in / (stored + 40*unique + 8*pointer + 8*delta)."""
    n_chunks = r.cuts.numel() - 1
    n_unique = int(r.uniq_ids.numel())
    n_delta = int((r.kind == 2).sum().item()) if r.kind is not None else 0
    lens = r.cuts[1:] - r.cuts[:-1]
    unique_bytes = int(lens[r.uniq_ids].sum().item()) if n_unique else 0
    stored = int(r.streams.numel()) if r.streams is not None else unique_bytes
    lsh_hits = int((r.base >= 0).sum().item()) if r.base is not None else 0
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
