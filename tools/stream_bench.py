"""Streaming ingest from pinned host memory: python tools/stream_bench.py [total MiB] [batch MiB] [profiles,comma-separated]
Reports the end-to-end rate with host->HBM copies overlapped, against the same data ingested in one resident call
(skipped above 16 GiB).  BASELINE.json configs[4] on one GPU: `tools/stream_bench.py 38144 1024 wikipedia,arxiv,news,code`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hmse_amd import IngestConfig, corpus, ingest, stream

tot = (int(sys.argv[1]) if len(sys.argv) > 1 else 4096) << 20
bat = (int(sys.argv[2]) if len(sys.argv) > 2 else 1024) << 20
profiles = (sys.argv[3] if len(sys.argv) > 3 else "wikipedia").split(",")
dev = torch.device("cuda:0")
cfg = IngestConfig()
GRAPH = os.environ.get("STREAM_GRAPH") == "1"
WINDOW = int(os.environ.get("STREAM_WINDOW_MIB", "0")) << 20   # > 0: only that many bytes of the stream stay resident (StreamIngest(window_bytes=), graph mode)
TAG = " [hipGraph chain]" if GRAPH else ""   # every batch = one enqueue of the device-count chain, replayed from a hipGraph
per = tot // len(profiles) // cfg.seg_size * cfg.seg_size
t0 = time.perf_counter()
host = torch.empty(per * len(profiles), dtype=torch.uint8).pin_memory()
for i, p in enumerate(profiles):
    host[i * per:(i + 1) * per] = torch.from_numpy(corpus.load(p, per, seed=42)[0])
tot = host.numel()
print(f"corpus: {len(profiles)} x {per / 1e9:.2f} GB ({', '.join(profiles)}) generated and pinned in {time.perf_counter() - t0:.1f} s", flush=True)
# warm-up outside the clock: module load, kernel attributes, allocator (a 16 GiB run's first iteration carries ~1.8 s of it)
w = stream.StreamIngest(cfg, 512 << 20, dev, graph=GRAPH)
for a in range(0, 512 << 20, 128 << 20):
    w.push(host[a: a + (128 << 20)])
w.finish(); del w
torch.cuda.synchronize()
iters = 2 if tot <= (16 << 30) else 1
for it in range(iters):
    torch.cuda.synchronize(); tc = time.perf_counter()
    st = stream.StreamIngest(cfg, tot, dev, graph=GRAPH, window_bytes=WINDOW or None)          # one-time: the resident corpus buffer (or window), the index arrays and tables
    torch.cuda.synchronize(); t0 = time.perf_counter()
    print(f"  index + corpus buffer allocated in {(t0 - tc) * 1e3:.0f} ms (outside the clock)", flush=True)
    per_batch = []
    for a in range(0, tot, bat):
        tb = time.perf_counter()
        st.push(host[a: a + bat])
        if os.environ.get("STREAM_BATCH_TIMES") == "1":   # serialises copy and compute: per-batch cost vs history
            torch.cuda.synchronize(); per_batch.append(round((time.perf_counter() - tb) * 1e3))
    res = st.finish()
    if per_batch:
        print("  ms per push (batch k's copy + batch k-1's kernels, synchronised):", per_batch, flush=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s = res.stats
    print(f"stream{TAG} iter {it}: {tot / 2**30:.1f} GiB in {-(-tot // bat)} batches, host->HBM included: {dt * 1e3:.0f} ms = {tot / dt / 2**30:.2f} GiB/s  "
          f"CF {tot / (s['stored_bytes'] + 40 * s['unique'] + 8 * s['pointer'] + 8 * s['delta']):.3f}  chunks {s['chunks']}  unique {s['unique']}  "
          f"delta {s['delta']}  HBM in use {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB" + (f"  resident window {WINDOW >> 20} MiB, {len(st.window_starts)} windows" if WINDOW else ""), flush=True)
    del st, res
if tot <= (16 << 30) and not WINDOW:
    for it in range(2):
        r = d = None
        torch.cuda.synchronize(); t0 = time.perf_counter()
        d = host.to(dev, non_blocking=True); torch.cuda.synchronize(); t1 = time.perf_counter()
        r = ingest.ingest_shard(d, cfg); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"single shot: copy {1e3 * (t1 - t0):.0f} ms + ingest {1e3 * (t2 - t1):.0f} ms = {tot / (t2 - t0) / 2**30:.2f} GiB/s incl. copy, {tot / (t2 - t1) / 2**30:.2f} GiB/s resident")
