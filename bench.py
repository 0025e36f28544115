#!/usr/bin/env python3
"""bench.py — end-to-end ingest GiB/s + CF of the L1-L4 hot path on N MI355X (one process per GPU).

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE from the env, RCCL backend).  One JSON line on
rank 0.  A "step" = one pass of the whole hot path (L2 FastCDC -> L3 SHA-256 + digest all-gather +
dedupe -> L4 MinHash/LSH -> L1 dictionary DEFLATE) over this rank's shard of the corpus, inputs
resident in HBM when the timed region starts.  Workload = BASELINE.json configs[2]/[3]: the full
pipeline over 10 GB of Wikipedia text (wiki-synth(seed=42) unless $HMSE_CORPUS_DIR has the real
sample), total size fixed as N grows ("strong" scaling, the way BASELINE.json's metric is worded).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"
_CLS = {8: "1024,10048,10048,true,false,false,false", 9: "1024,22976,22976,true,true,true,true", 10: "1024,32768,32768,true,true,true,true",
        11: "1024,65536,32768,false,false,false,false", 12: "1024,13952,13952,true,true,false,false", 13: "1024,17408,17408,true,true,true,false"}
STAGE_NAMES = {2: "l2_hash_kernel", 3: "l3_sha256_kernel", 5: "l4_minhash_kernel",
               14: "l1_encode_kernel<0,12288>", 15: "l1_encode_kernel<12288,32768>"}
# match kernels: slots 8..13 = plain jobs of a size class, 18..23 = its dictionary jobs (last template argument)
STAGE_NAMES.update({slot: f"l1_deflate_kernel<{args},false>" for slot, args in _CLS.items()})
STAGE_NAMES.update({slot + 10: f"l1_deflate_kernel<{args},true>" for slot, args in _CLS.items()})
# the encode kernels run twice per call: DELTA records (after the dictionary jobs; slots 30, 31), then FULL records (14, 15)
STAGE_NAMES.update({30: "l1_encode_kernel<0,12288> [DELTA records]", 31: "l1_encode_kernel<12288,32768> [DELTA records]"})
# what the SQ counters say about the kernels that can be "dominant" (profiles/r1/h_pmc_sq_counters_2GB.csv)
VALU_NOTE = {"l4_minhash_kernel": "; SQ counters: SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES x 8 waves per SIMD = 1.1, i.e. the vector ALUs are "
                                  "saturated (9.25 instructions per (distinct shingle, seed) pair)"}
# match-kernel size classes S, SG2, SG3, B, S2, SG (hmse_amd/csrc/l1_deflate.hip) -> profile slot
DEFLATE_CLASS_SLOT = {0: 8, 1: 9, 2: 10, 3: 11, 4: 12, 5: 13}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--bytes", type=float, default=10e9, help="total corpus bytes over all ranks (default 10 GB)")
    ap.add_argument("--corpus", default="wikipedia")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--layers", default="full", help="ablation preset (hmse_amd.config.ABLATIONS)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--global-l4", action="store_true",
                    help="N > 1: base selection over all shards (signature all-gather + cross-GPU base fetch) instead of shard-local; "
                         "the stored bytes are then those of the 1-GPU run")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the side measurements of BASELINE configs[1] / configs[4] (one rank)")
    ap.add_argument("--cpu-sample-mib", type=int, default=0, help="CPU baseline sample (0 = 4 MiB x 2 x cores, <= 256 MiB)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------ CPU leg
def _cpu_segment(args):
    """The reference's CPU pipeline over one 4 MiB segment (runs in a worker process), same scope as one GPU shard of that
    segment: oracle FastCDC -> hashlib.sha256 -> first-occurrence dedupe -> oracle MinHash/LSH on the unique chunks ->
    zlib level 9 (raw, zdict = LSH base) per unique chunk."""
    import hashlib
    import zlib
    seg, cfg_kw = args
    from oracle import oracle as O
    cfg = O.default_cfg(**cfg_kw)
    cuts = O.cdc(seg, cfg)
    n = len(cuts) - 1
    digs = np.frombuffer(b"".join(hashlib.sha256(seg[int(cuts[i]):int(cuts[i + 1])]).digest() for i in range(n)), np.uint8).reshape(n, 32)
    fo, _ = O.dedup(digs)
    uniq = np.nonzero(fo == np.arange(n))[0].astype(np.uint64)
    sig = O.minhash_chunks(seg, cuts, cfg, uniq)
    _, base = O.lsh(sig, cfg)
    stored = 0
    for k, c in enumerate(uniq):
        ch = seg[int(cuts[c]):int(cuts[c + 1])].tobytes()
        b = int(uniq[base[k]]) if base[k] >= 0 else -1
        if b >= 0:
            co = zlib.compressobj(9, zlib.DEFLATED, -15, 9, zlib.Z_DEFAULT_STRATEGY, seg[int(cuts[b]):int(cuts[b + 1])].tobytes())
            delta = len(co.compress(ch) + co.flush())
        co = zlib.compressobj(9, zlib.DEFLATED, -15, 9, zlib.Z_DEFAULT_STRATEGY)
        full = len(co.compress(ch) + co.flush())
        # rule 7 (README.md:1328, 2175): a delta of at most a fifth of the chunk is kept, a larger one iff it nets savings
        stored += delta if b >= 0 and (5 * delta <= len(ch) or delta + 8 < full) else full
    return n, stored, cuts, uniq, base


def _cpu_segment_oracle_deflate(args):
    """Untimed second pass: the build's own encoder definition (oracle DEFLATE) over the same chunks and bases."""
    seg, cfg_kw, cuts, uniq, base = args
    from oracle import oracle as O
    _, off, _ = O.deflate_chunks(seg, cuts, O.default_cfg(**cfg_kw), uniq, base)
    return int(off[-1])


def usable_cores() -> int:
    """Host cores this process may actually use: the scheduler affinity and the cgroup CPU quota count, not just os.cpu_count()
    (a one-GPU box of the pool shows all 256 cores of its host but grants a share of them)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(host: np.ndarray, sample_mib: int) -> dict:
    """The Python zlib/hashlib + oracle CPU path on a bounded sample: all host cores, and one core.
    Runs BEFORE the process touches the GPU (it forks workers)."""
    import multiprocessing as mp
    from oracle import oracle as O
    O.build()
    cores = usable_cores()
    seg = 4 << 20
    # every usable host core gets work: eight 4 MiB segments per core (bounded at 2 GiB), ~1.3 s of one-core work each
    nseg = sample_mib // 4 if sample_mib else min(8 * cores, 512)
    nseg = max(1, min(nseg, host.size // seg))
    segs = [(host[i * seg:(i + 1) * seg], {}) for i in range(nseg)]
    workers = min(cores, nseg)
    t0 = time.time()
    _cpu_segment(segs[0])
    dt1 = time.time() - t0
    with mp.get_context("fork").Pool(workers) as pool:
        t0 = time.time()
        res = pool.map(_cpu_segment, segs)
        dt = time.time() - t0
        stored_orc = sum(pool.map(_cpu_segment_oracle_deflate, [(s[0], s[1], r[2], r[3], r[4]) for s, r in zip(segs, res)]))
    nbytes = nseg * seg
    return {"value": nbytes / dt / 2**30, "unit": "GiB/s", "cores": workers, "host_cores": os.cpu_count() or 1, "usable_cores": cores, "kind": "port",
            "one_core_GiB_per_s": seg / dt1 / 2**30,
            "sample": f"first {nseg} x 4 MiB segments of the same corpus ({nbytes >> 20} MiB), each segment its own dedupe/LSH scope: oracle "
                      f"FastCDC + hashlib.sha256 + dedupe + oracle MinHash/LSH + zlib level 9 (raw, zdict = LSH base, delta rule), one "
                      f"segment per worker; one_core = the first segment alone",
            "seconds": round(dt, 2), "sample_bytes": nbytes, "n_segments": nseg,
            "cf_payload_zlib9_same_sample": nbytes / max(1, sum(r[1] for r in res)),
            "cf_payload_oracle_deflate_same_sample": nbytes / max(1, stored_orc), "stored_bytes_oracle_deflate": stored_orc}


def measured_traffic(kernel: str, total_bytes: int, world: int):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE doubled per the gfx950
    note of MI355X_MICROARCH.md, + WRITE_SIZE), valid only for the workload they were taken on; else None."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(path))
        e = t["kernels"][kernel]
        if world == 1 and t["total_bytes"] == total_bytes:
            return {"hbm_bytes": e["hbm_bytes"], "GB": round(e["hbm_bytes"] / 1e9, 3), "source": t["source"]}
    except (OSError, KeyError, ValueError):
        pass
    return None


def valu_issue_note():
    """VALU-busy fraction per kernel family from the committed SQ counter passes (profiles/sq_valu.json, written by
    tools/make_sq.py from rocprofv3 --pmc runs): SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES per SIMD."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "sq_valu.json")))
    except (OSError, ValueError):
        return None


def other_configs(data, host, dev, cfg_full) -> dict:
    """BASELINE.json configs[1] and the one-rank form of configs[4], measured in the driver's own run AFTER the headline's timed region
    (both existed only as builder-run files).  Outside `value`; reported under "other_configs"."""
    import torch
    from hmse_amd import ABLATIONS, IngestConfig, corpus, ingest, ops, stream
    out = {}
    seg = cfg_full.seg_size
    # configs[1]: FastCDC (8 KiB avg) + SHA-256 dedup on 1 GB, cut points / digests bit-exact vs the CPU (tests/test_gpu_parity.py)
    n1 = min(int(1e9), data.numel()) // seg * seg
    cfg1 = IngestConfig(layers=ABLATIONS["cdc_dedupe"])
    d1 = data[:n1]
    so1 = ops.segment_offsets(n1, seg, dev)
    r1 = None
    for _ in range(3):
        r1 = ingest.ingest_shard(d1, cfg1, so1, want_stats=False)
    torch.cuda.synchronize()
    steps1 = 20
    t0 = time.perf_counter()
    for _ in range(steps1):
        r1 = None
        r1 = ingest.ingest_shard(d1, cfg1, so1, want_stats=False)
    torch.cuda.synchronize()
    dt1 = (time.perf_counter() - t0) / steps1
    st1 = ingest.shard_stats(r1)
    out["configs[1]"] = {"workload": f"FastCDC 2/8/32 KiB + SHA-256 dedupe over {n1 / 1e9:.2f} GB (the first bytes of the headline corpus), resident in HBM",
                         "GiB_per_s": round(n1 / dt1 / 2**30, 2), "ms_per_step": round(dt1 * 1e3, 3), "steps": steps1,
                         "frac_hbm_read_roofline": round(n1 / dt1 / 1e9 / HBM_PEAK_GBPS, 5), "chunks": st1["chunks"],
                         "unique_chunk_ratio": round(st1["unique"] / max(1, st1["chunks"]), 4)}
    del r1, d1
    # configs[4], one rank: mixed corpora streamed from pinned host memory, every batch ONE replay of the hipGraph-captured chain,
    # host -> HBM copies overlapped and INCLUDED (the reference's batch loop, README.md:1519-1580)
    profiles = ("wikipedia", "arxiv", "news", "code")
    per = int(os.environ.get("HMSE_BENCH_STREAM_MIB", "2048")) << 20
    per = min(per, host.size) // seg * seg
    bat = min(1 << 30, per)
    tg = time.perf_counter()
    pinned = torch.empty(per * len(profiles), dtype=torch.uint8).pin_memory()
    for i, pr in enumerate(profiles):
        pinned[i * per:(i + 1) * per] = torch.from_numpy(host[:per] if pr == "wikipedia" else corpus.load(pr, per, seed=42)[0])
    t_gen = time.perf_counter() - tg
    tot = pinned.numel()
    w = stream.StreamIngest(cfg_full, 2 * bat, dev, graph=True)      # warm-up: capture of the batch size, allocator, module load
    w.push(pinned[:bat]); w.push(pinned[bat: 2 * bat]); w.finish(); del w
    torch.cuda.synchronize()
    st = stream.StreamIngest(cfg_full, tot, dev, graph=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a0 in range(0, tot, bat):
        st.push(pinned[a0: a0 + bat])
    res = st.finish()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s = res.stats
    nb = -(-tot // bat)
    out["configs[4] on one rank"] = {
        "workload": f"{len(profiles)} x {per / 2**30:.2f} GiB ({', '.join(profiles)}; wiki-synth profiles, seed 42) streamed from pinned host memory in {nb} batches of "
                    f"{bat >> 20} MiB, one hipGraph replay of hmse_stream_batch per batch, host -> HBM copies overlapped and included",
        "GiB_per_s": round(tot / dt / 2**30, 2), "ms_per_batch": round(dt / nb * 1e3, 2), "batches": nb, "bytes": tot,
        "cf": round(tot / max(1, s["stored_bytes"] + 40 * s["unique"] + 8 * s["pointer"] + 8 * s["delta"]), 4), "chunks": s["chunks"],
        "delta_records": s["delta"], "frac_hbm_read_roofline": round(tot / dt / 1e9 / HBM_PEAK_GBPS, 6), "corpus_gen_and_pin_s": round(t_gen, 2)}
    return out


# ------------------------------------------------------------------------------------------ main
def launch_ranks(a) -> int:
    """`python bench.py --gpus N` started WITHOUT torch.distributed.run (no RANK / WORLD_SIZE in the environment): start the N ranks
    here — a child `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>`, spawned before this
    process imports torch or touches a GPU — and relay rank 0's JSON line.  Never a 1-GPU run labelled N."""
    import socket
    import subprocess
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {a.gpus} without a launcher: starting {a.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    lines = [ln for ln in p.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not lines:
        print(f"[bench] the {a.gpus}-rank run failed (rc {p.returncode}); no result line", file=sys.stderr, flush=True)
        return p.returncode or 1
    sys.stdout.write(lines[-1] + "\n"); sys.stdout.flush()
    return 0


def one_gpu_cf_reference(corpus_name: str, total_bytes: int):
    """CF of the ONE-GPU run of the same corpus and size (north_star: "CF identical"): an N > 1 line reports it beside its own CF,
    whose shard-local L4 finds fewer bases.  From the committed 1-GPU bench lines (profiles/cf_one_gpu.json)."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "cf_one_gpu.json")))
        e = t[corpus_name][str(total_bytes)]
        return {"cf": e["cf"], "cf_payload": e.get("cf_payload"), "source": e["source"]}
    except (OSError, KeyError, ValueError):
        return None


def main():
    a = parse()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if a.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(a))
    # stdout carries exactly ONE line (the JSON): libraries that chat on fd 1 (RCCL prints a version banner there at its
    # first collective) are sent to stderr for the whole run, and the result is written to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    from hmse_amd import ABLATIONS, IngestConfig, _lib, corpus, ingest, ops

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: the result line would name the wrong GPU count")
    if os.environ.get("HMSE_BENCH_LAUNCH_ONLY") == "1":
        # launcher check (tests/test_host.py, no GPU): every rank reports in over gloo, rank 0 prints who came
        if world > 1:
            dist.init_process_group("gloo")
            seen = [None] * world
            dist.all_gather_object(seen, (rank, local_rank))
            dist.destroy_process_group()
        else:
            seen = [(rank, local_rank)]
        if rank == 0:
            os.write(json_fd, (json.dumps({"launch_check": True, "n_gpus": world, "ranks": sorted(r for r, _ in seen)}) + "\n").encode())
        return
    cfg = IngestConfig(layers=ABLATIONS[a.layers])
    seg = cfg.seg_size
    total = int(a.bytes) if a.scaling == "strong" else int(a.bytes) * world
    n_seg_total = max(world, total // seg)
    # contiguous runs of whole segments per rank
    s0 = rank * n_seg_total // world
    s1 = (rank + 1) * n_seg_total // world
    n_local = (s1 - s0) * seg
    t0 = time.time()
    host, source = corpus.load(a.corpus, n_local, first_byte=s0 * seg)
    t_gen = time.time() - t0
    # CPU baseline leg first: it forks worker processes, which must happen before this process initialises the GPU
    cpu = None
    if not a.no_cpu_baseline and world == 1 and a.layers == "full":
        try:
            cpu = cpu_baseline(host, a.cpu_sample_mib)
        except Exception as e:  # noqa: BLE001 — the baseline leg must not lose the GPU measurement
            cpu = {"error": repr(e)}
    # HMSE_BENCH_REHEARSE=1: every rank on GPU 0, exchanges over gloo — the N-rank control flow (sharding, the digest
    # exchange, max-over-ranks timing, the rank-0 JSON line) on a one-GPU box; never a measurement
    rehearse = os.environ.get("HMSE_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # HMSE_BENCH_FORCE_DIST=1 exercises the RCCL path (init, all-gather, reductions) even at world size 1
    distributed = world > 1 or os.environ.get("HMSE_BENCH_FORCE_DIST") == "1"
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1 and "RANK" not in os.environ:   # HMSE_BENCH_FORCE_DIST=1 under plain `python bench.py`: a rendezvous of one
            import socket
            sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    t0 = time.time()
    data = torch.from_numpy(host).to(dev)
    torch.cuda.synchronize()
    t_h2d = time.time() - t0
    seg_off = ops.segment_offsets(n_local, seg, dev)
    lib = _lib.hip_lib()

    def step():
        return ingest.ingest_shard(data, cfg, seg_off, distributed=distributed, want_stats=False, global_l4=a.global_l4 and distributed)

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    res = None
    for _ in range(a.warmup):
        res = None  # free the previous step's outputs first: the caching allocator then reuses them
        res = step()
    lib.hmse_profile_enable(1)
    for s in STAGE_NAMES:
        lib.hmse_profile_read(s, None, None, 1)
    barrier()
    t0 = time.perf_counter()
    step_marks = []      # host clock after every step's return (no extra synchronisation: a step ends in its own read-backs)
    for _ in range(a.steps):
        res = None
        res = step()
        step_marks.append(time.perf_counter())
    barrier()
    dt = time.perf_counter() - t0
    step_ms_host = [round((b - a_) * 1e3, 1) for a_, b in zip([t0] + step_marks[:-1], step_marks)]
    lib.hmse_profile_enable(0)
    if distributed:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # per-kernel durations of the timed steps (HIP events on the launch stream, inside the C-ABI)
    kern = {}
    tokens = {}          # tokens written (match kernels) / read (encode kernels) per launch, counted on the device
    for s, name in STAGE_NAMES.items():
        ms, n, tk = C.c_double(), C.c_uint64(), C.c_uint64()
        lib.hmse_profile_read(s, C.byref(ms), C.byref(n), 1)
        lib.hmse_profile_counter(s, C.byref(tk), 1)
        if n.value:
            kern[name] = {"avg_ms": ms.value / n.value, "launches": int(n.value)}
            tokens[name] = tk.value / n.value
    st = ingest.shard_stats(res)
    # algorithmic bytes per launch: L2/L3 read every input byte once; L4a reads the unique
    # bytes; L1 reads unique + dictionary bytes and writes the streams
    lens = res.cuts[1:] - res.cuts[:-1]
    dict_bytes = 0
    if res.base is not None and res.uniq_ids.numel():
        hb = res.base >= 0
        dict_bytes = int(lens[res.uniq_ids[res.base[hb]]].sum().item()) if bool(hb.any()) else 0
    alg = {"l2_hash_kernel": n_local, "l3_sha256_kernel": n_local + 32 * st["chunks"],
           "l4_minhash_kernel": st["unique_bytes"] + 512 * st["unique"]}
    # DEFLATE jobs per size class: window bytes read (chunk + dictionary) + the share of stream bytes written
    if res.kind is not None and res.uniq_ids.numel():
        ul = lens[res.uniq_ids]
        hb = res.base >= 0 if res.base is not None else torch.zeros_like(ul, dtype=torch.bool)
        dl = torch.clamp(ul[res.base.clamp(min=0)], max=32768) if res.base is not None else torch.zeros_like(ul)
        # match jobs: one per chunk — its own window, or chunk + dictionary for a chunk with a base (that job emits both the
        # FULL and the DELTA token list); encode jobs: one per record

        def by_class(T, Ltok, off):
            c_s, c_s2, c_sg, c_sg2, c_sg3 = ops.DEFLATE_CLASS_CAPS
            cls = torch.where(T <= c_s, 0, torch.where(T <= c_s2, 4, torch.where(T <= c_sg, 5, torch.where(T <= c_sg2, 1, torch.where(T <= c_sg3, 2, 3)))))
            for c, slot in DEFLATE_CLASS_SLOT.items():
                m = cls == c
                # match kernel: window read + token list written (4 B per token, COUNTED by the kernel: hmse_profile_counter)
                alg[STAGE_NAMES[slot + off]] = int(T[m].sum().item() + 4 * tokens.get(STAGE_NAMES[slot + off], 0))

        by_class(ul[~hb], ul[~hb], 0)                # plain jobs
        by_class((ul + dl)[hb], ul[hb], 10)          # dictionary jobs (one token list: the DELTA record's)
        enc_slots = (14, 15, 30, 31)
        tot_tok = sum(tokens.get(STAGE_NAMES[sl], 0) for sl in enc_slots) or 1
        for slot in enc_slots:
            # encode kernel: token list read (counted), stream written (share of the stored bytes by token share)
            alg[STAGE_NAMES[slot]] = int(4 * tokens.get(STAGE_NAMES[slot], 0) + st["stored_bytes"] * tokens.get(STAGE_NAMES[slot], 0) / tot_tok)
    stage_roof = {}
    for name, k in kern.items():
        if name not in alg:
            continue
        gbps = alg[name] / (k["avg_ms"] * 1e-3) / 1e9
        stage_roof[name] = {"avg_ms": round(k["avg_ms"], 4), "alg_bytes": int(alg[name]), "GBps": round(gbps, 2),
                            "frac_hbm": round(gbps / HBM_PEAK_GBPS, 5)}
    dom = max(stage_roof, key=lambda nme: stage_roof[nme]["avg_ms"]) if stage_roof else None

    # read path, outside the timed region: every stored record is inflated on the GPU and its SHA-256
    # re-checked; on one GPU the whole corpus is also reassembled (pointers included) and compared byte for byte
    read_info = None
    remote_dicts = a.global_l4 and distributed   # records may then name dictionaries stored on other ranks: no per-rank read-back
    if remote_dicts:
        read_info = {"scope": "skipped: with --global-l4 a record's dictionary may be stored on another rank (read.read_store decodes the merged store, read.reconstruct_shards the shard results)"}
    elif res.streams is not None and res.digests is not None and os.environ.get("HMSE_BENCH_NO_VERIFY") != "1":
        from hmse_amd import read
        for s in (16, 17):
            lib.hmse_profile_read(s, None, None, 1)
        lib.hmse_profile_enable(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if world == 1 and not distributed:
            back = read.reconstruct_shard(res, verify=True)
            torch.cuda.synchronize()
            t_read = time.perf_counter() - t0
            identical = bool(torch.equal(back, data))
            del back
            read_info = {"scope": "whole shard: inflate + pointer assembly + SHA-256 of every chunk", "identical_to_input": identical,
                         "sha256_verified_chunks": st["chunks"], "GiB_per_s": round(n_local / t_read / 2**30, 3), "ms": round(t_read * 1e3, 2)}
        else:
            n_ok = read.verify_stored(res)
            torch.cuda.synchronize()
            t_read = time.perf_counter() - t0
            read_info = {"scope": "stored records of rank 0: inflate + SHA-256", "sha256_verified_chunks": n_ok, "ms": round(t_read * 1e3, 2)}
        lib.hmse_profile_enable(0)
        for s, nm in ((16, "l1_inflate_kernel_ms"), (17, "assemble_kernel_ms")):
            ms, n = C.c_double(), C.c_uint64()
            lib.hmse_profile_read(s, C.byref(ms), C.byref(n), 1)
            if n.value:
                read_info[nm] = round(ms.value / n.value, 3)

    # the product's output: the shard's manifest records, packed on the GPU (outside the timed region, reported beside it)
    manifest_info = None
    if res.streams is not None and os.environ.get("HMSE_BENCH_NO_MANIFEST") != "1":
        from hmse_amd import manifest
        sb = res.shard_bases
        manifest.pack_manifest_device(res, rank, world if sb else 1)   # warm-up (allocator, first launch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        unit, blob, index, cmap, ptrs = manifest.pack_manifest_device(res, rank, world if sb else 1)
        torch.cuda.synchronize()
        t_pack = time.perf_counter() - t0
        manifest_info = {"pack_ms": round(t_pack * 1e3, 2), "blob_bytes": int(blob.numel()), "index_entries": int(index.shape[0]),
                         "map_entries": int(cmap.shape[0]), "pointer_records": int(ptrs.shape[0]), "lba_unit": unit,
                         "scope": "hmse_manifest_pack + record/pointer prefix sums, records resident in HBM when the clock stops (rank 0)"}
        del blob, index, cmap, ptrs
    other = None
    if world == 1 and not distributed and a.layers == "full" and os.environ.get("HMSE_BENCH_NO_OTHER") != "1" and not a.no_other_configs:
        try:
            other = other_configs(data, host, dev, cfg)
        except Exception as e:  # noqa: BLE001 — a side measurement must not lose the headline
            other = {"error": repr(e)}
    stats = [st]
    if distributed:
        allst = [None] * world
        dist.all_gather_object(allst, st)
        stats = allst
    if rank == 0:
        tot = ingest.merge_stats(stats)
        value = tot["bytes"] * a.steps / dt / 2**30
        out = {
            "metric": "ingest_GiB_per_s", "value": round(value, 3), "unit": "GiB/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "step_ms_host": step_ms_host, "higher_is_better": True, "scaling": a.scaling,
            "vs_baseline": None, "dtype": "u8", "data": f"synthetic: {source}" if source.startswith("wiki") else source,
            "config": {"workload": f"{a.layers} ingest (" + " + ".join(nm for bit, nm in (
                           (2, "FastCDC 2/8/32 KiB"), (4, "SHA-256 dedupe"), (8, "MinHash-128/LSH 4x32"),
                           (1, "dictionary DEFLATE level-9 profile" if cfg.layers & 8 else "DEFLATE level-9 profile")) if cfg.layers & bit)
                                   + f") over {tot['bytes'] / 1e9:.2f} GB {a.corpus}",
                       "total_bytes": tot["bytes"], "seg_size": seg, "sharding": f"{world} x contiguous 4 MiB-segment runs",
                       "collective": ("all_gather(digests) over gloo — REHEARSAL, all ranks on one GPU, not a measurement" if rehearse else ("all_gather(digests) + all_gather(signatures) + all_to_all(base chunks) over RCCL" if a.global_l4 else "all_gather(digests) over RCCL")) if distributed else "none"},
            "cf": round(tot["cf"], 4), "cf_payload": round(tot["cf_payload"], 4), "unique_chunk_ratio": round(tot["unique_chunk_ratio"], 4),
            "lsh_hit_rate": round(tot["lsh_hit_rate"], 4), "delta_rate": round(tot["delta_rate"], 4), "chunks": tot["chunks"],
            "frac_hbm_read_roofline": round(tot["bytes"] * a.steps / dt / 1e9 / (HBM_PEAK_GBPS * world), 6),
            "h2d_inclusive_GiB_per_s": round(tot["bytes"] / (dt / a.steps + t_h2d) / 2**30, 3),
            "corpus_gen_s": round(t_gen, 2), "h2d_s": round(t_h2d, 2),
        }
        if world > 1:
            # the CF claim of north_star ("CF identical to the CPU reference" = the one-GPU run's) made visible: shard-local L4 finds
            # fewer bases as the shards shrink; --global-l4 restores the one-GPU bytes (tests/test_gpu_ingest.py)
            out["cf_one_gpu_reference"] = one_gpu_cf_reference(a.corpus, tot["bytes"])
            out["l4_scope"] = "global (signature all-gather + cross-GPU base fetch)" if a.global_l4 else "shard-local"
        if dom:
            r = stage_roof[dom]
            # stages = groups of kernels: the DEFLATE match kernels run as up to twelve launches (size class x plain/dictionary)
            groups = {"L1 DEFLATE match (l1_deflate_kernel, all size classes)": "l1_deflate_kernel", "L1 DEFLATE encode (l1_encode_kernel)": "l1_encode_kernel",
                      "L4 MinHash (l4_minhash_kernel)": "l4_minhash_kernel", "L3 SHA-256 (l3_sha256_kernel)": "l3_sha256_kernel", "L2 Gear hash (l2_hash_kernel)": "l2_hash_kernel"}
            stage_ms = {g: sum(v["avg_ms"] for k, v in stage_roof.items() if k.startswith(pre)) for g, pre in groups.items()}
            stage_alg = {g: sum(v["alg_bytes"] for k, v in stage_roof.items() if k.startswith(pre)) for g, pre in groups.items()}
            dstage = max(stage_ms, key=stage_ms.get)
            out["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": r["GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                               "frac": r["frac_hbm"], "traffic": measured_traffic(dom, tot["bytes"], world),
                               "dominant_stage": {"stage": dstage, "ms_per_step": round(stage_ms[dstage], 2),
                                                  "share_of_step": round(stage_ms[dstage] / (dt / a.steps * 1e3), 3),
                                                  "achieved_GBps": round(stage_alg[dstage] / max(stage_ms[dstage], 1e-9) / 1e6, 2),
                                                  "frac_hbm": round(stage_alg[dstage] / max(stage_ms[dstage], 1e-9) / 1e6 / HBM_PEAK_GBPS, 5)},
                               "valu_issue": valu_issue_note(),
                               "note": "`kernel` = the longest single launch, `dominant_stage` = the longest group of launches; both are "
                                       "integer-VALU/LDS-bound and priced against the HBM roof (SURVEY.md §8d); rank 0" + VALU_NOTE.get(dom, "")}
        if manifest_info:
            out["manifest"] = manifest_info
            out["manifest_inclusive_GiB_per_s"] = round(tot["bytes"] / (dt / a.steps + manifest_info["pack_ms"] * 1e-3) / 2**30, 3)
        if other:
            out["other_configs"] = other
        out["stage_roofline"] = stage_roof
        if read_info:
            out["read_path"] = read_info
        if cpu is not None:
            if "n_segments" in cpu:
                # the GPU path over the SAME sample and scope as the CPU leg (every 4 MiB segment its own shard): its stored
                # bytes must equal the oracle encoder's exactly (bit-exact streams), zlib-9's CF is the reference point
                stored_gpu = 0
                for i in range(cpu["n_segments"]):
                    r = ingest.ingest_shard(data[i * seg:(i + 1) * seg], cfg, want_stats=True)
                    stored_gpu += r.stats["stored_bytes"]
                cpu["cf_payload_gpu_same_sample"] = cpu["sample_bytes"] / max(1, stored_gpu)
                cpu["gpu_stored_bytes_equal_oracle_deflate"] = bool(stored_gpu == cpu.pop("stored_bytes_oracle_deflate"))
            out["cpu_baseline"] = cpu
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
