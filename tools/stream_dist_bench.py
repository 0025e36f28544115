"""BASELINE.json configs[4] — "4 x 10 GB mixed corpora (Wikipedia/arXiv/news/code) streamed, 8 x MI355X, hipGraph-captured per-batch
pipeline" — as an N-rank job, one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tools/stream_dist_bench.py --gpus N [--total-mib 38144] [--piece-mib 1024] [--profiles wikipedia,arxiv,news,code]

The launcher starts the ranks BEFORE anything touches a GPU (torch.distributed.run spawns fresh interpreters).  The logical stream
is the concatenation of the profiles; global batch b = N pieces of --piece-mib, rank r takes piece r (hmse_amd/stream_dist.py).
Per batch: phase A (hipGraph) -> all-gather of the exchange rows over RCCL -> phase B (hipGraph); host -> HBM copies of the next
piece overlap.  Rank 0 prints ONE JSON line: whole-job GiB/s (max over ranks of the wall time, host -> HBM included), CF, counts.

HMSE_BENCH_REHEARSE=1: every rank on GPU 0 and the exchange over gloo — the control flow of an N-rank run on a one-GPU box (two
real processes, real kernels, the real per-batch exchange; everything except RCCL itself).  Never a measurement.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--total-mib", type=int, default=4096, help="logical stream size over all ranks")
    ap.add_argument("--piece-mib", type=int, default=1024, help="nominal piece (per rank and batch)")
    ap.add_argument("--profiles", default="wikipedia,arxiv,news,code")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--verify", action="store_true", help="decode this rank's stored records and re-check their SHA-256 (outside the clock)")
    ap.add_argument("--global-l4", action="store_true", help="base selection over all ranks (stream_dist.GlobalL4StreamIngest: signatures all-gathered "
                    "per batch, remote dictionaries fetched by all-to-alls; enqueued stage by stage, no hipGraph); needs --gpus > 1")
    ap.add_argument("--global-l4-graph", action="store_true", help="global L4 with every phase hipGraph-captured (stream_gl4.GraphGlobalL4StreamIngest: "
                    "fixed-size digest and signature rows all-gathered, only the remote-dictionary fetch eager); also at --gpus 1")
    ap.add_argument("--always-exchange", action="store_true", help="--gpus 1: run the per-batch collectives on RCCL anyway (what one GPU can show of their cost)")
    a = ap.parse_args()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    from hmse_amd import IngestConfig, corpus, ingest, read, stream_dist, stream_gl4

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.global_l4 and world < 2:
        raise SystemExit("--global-l4 is a multi-rank mode (one rank: the default stream already selects bases over everything)")
    rehearse = os.environ.get("HMSE_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    cfg = IngestConfig()
    seg = cfg.seg_size
    piece = (a.piece_mib << 20) // seg * seg
    profiles = a.profiles.split(",")
    n_batches = max(1, (a.total_mib << 20) // (piece * world))
    total = n_batches * piece * world
    per_profile = -(-n_batches // len(profiles))          # whole batches per profile, profiles in order
    # this rank's pieces: piece (b, rank) of the logical stream = bytes [(b * world + rank) * piece, + piece) of it; the stream is
    # profile p for batches [p * per_profile, (p + 1) * per_profile)
    t0 = time.time()
    host = torch.empty(n_batches * piece, dtype=torch.uint8).pin_memory()
    for b in range(n_batches):
        p = profiles[min(b // per_profile, len(profiles) - 1)]
        first = ((b % per_profile) * world + rank) * piece
        host[b * piece: (b + 1) * piece] = torch.from_numpy(corpus.load(p, piece, first_byte=first, seed=42)[0])
    t_gen = time.time() - t0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or a.always_exchange:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1 and "RANK" not in os.environ:
            import socket
            sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo" if rehearse else "nccl", **({} if rehearse else {"device_id": dev}))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up outside the clock: module load, kernel attributes, allocator, the collective's first call
    wp = min(64 << 20, piece)
    wn = max(1, min(4, host.numel() // wp))
    mk = (lambda cap, pc: stream_gl4.GraphGlobalL4StreamIngest(cfg, cap, pc, dev, world, rank, graph=not a.no_graph, always_exchange=a.always_exchange)) if a.global_l4_graph else \
        (lambda cap, pc: stream_dist.GlobalL4StreamIngest(cfg, cap, pc, dev, world, rank)) if a.global_l4 else \
        (lambda cap, pc: stream_dist.DistStreamIngest(cfg, cap, pc, dev, world, rank, graph=not a.no_graph, always_exchange=a.always_exchange))
    w = mk(wn * wp, wp)
    for k in range(wn):
        w.push(host[k * wp: (k + 1) * wp])
    w.finish(); del w
    barrier()
    s = mk(n_batches * piece, piece)
    barrier()
    t0 = time.perf_counter()
    for b in range(n_batches):
        s.push(host[b * piece: (b + 1) * piece])
    res = s.finish()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    st = res.stats
    verified = read.verify_stored(res) if a.verify and not (a.global_l4 or (a.global_l4_graph and world > 1)) else None    # (global L4: dictionaries on other ranks — tests/ decode all ranks together)
    stats = [st]
    if world > 1:
        stats = [None] * world
        dist.all_gather_object(stats, st)
    if rank == 0:
        tot = ingest.merge_stats(stats)
        out = {"metric": "stream_ingest_GiB_per_s", "value": round(total / dt / 2**30, 3), "unit": "GiB/s", "n_gpus": world, "higher_is_better": True,
               "scaling": "strong", "dtype": "u8", "data": "synthetic: wiki-synth profiles " + ",".join(profiles),
               "config": {"workload": f"{total / 1e9:.2f} GB mixed corpora streamed in {n_batches} global batches of {world} x {piece >> 20} MiB pieces, full L1-L4, "
                                      f"{'hipGraph-captured' if not (a.no_graph or a.global_l4) else 'eagerly enqueued'} per-batch chain, host->HBM copies included",
                          "collective": ("all_gather(digests) + all_gather(signatures) + 3 all_to_all (remote dictionaries) per batch over " if a.global_l4 else "all_gather(exchange rows) per batch over ") + ("gloo — REHEARSAL, all ranks on one GPU, not a measurement" if rehearse else "RCCL") if world > 1 else "none",
                          "global_chunk_order": "(batch, rank, local)", "l4_scope": "global, captured phases (stream_gl4)" if a.global_l4_graph else "global" if a.global_l4 else "rank-local",
                          "collectives_at_world_1": "run on RCCL anyway (--always-exchange)" if (world == 1 and a.always_exchange) else None},
               "ms_total": round(dt * 1e3, 1), "ms_per_batch": round(dt * 1e3 / n_batches, 2), "cf": round(tot["cf"], 4), "cf_payload": round(tot["cf_payload"], 4),
               "chunks": tot["chunks"], "unique_chunk_ratio": round(tot["unique_chunk_ratio"], 4), "delta_rate": round(tot["delta_rate"], 4),
               "n_global_chunks": res.n_global, "corpus_gen_s": round(t_gen, 1), "hbm_in_use_GiB_rank0": round(torch.cuda.max_memory_allocated() / 2**30, 1)}
        if verified is not None:
            out["sha256_verified_records_rank0"] = verified
        if a.global_l4 or a.global_l4_graph:
            out["remote_dictionaries_rank0"] = s.remote_dictionaries
            out["ghost_bytes_fetched_rank0"] = s.ghost_bytes_fetched
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or a.always_exchange:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
