"""Streaming ingest from pinned host memory: python tools/stream_bench.py [total MiB] [batch MiB].
Reports the end-to-end rate with host->HBM copies overlapped against the same data ingested in one resident call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hmse_amd import IngestConfig, corpus, ingest, stream

tot = (int(sys.argv[1]) if len(sys.argv) > 1 else 4096) << 20
bat = (int(sys.argv[2]) if len(sys.argv) > 2 else 1024) << 20
dev = torch.device("cuda:0")
cfg = IngestConfig()
host = torch.from_numpy(corpus.wiki_synth(tot, seed=42)).pin_memory()
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st = stream.StreamIngest(cfg, tot, dev)
    for a in range(0, tot, bat):
        st.push(host[a: a + bat])
    res = st.finish()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"stream iter {it}: {tot / 2**30:.1f} GiB in {tot // bat} batches, host->HBM included: {dt * 1e3:.0f} ms = {tot / dt / 2**30:.2f} GiB/s  "
          f"CF {tot / (res.stats['stored_bytes'] + 40 * res.stats['unique'] + 8 * res.stats['pointer'] + 8 * res.stats['delta']):.3f}", flush=True)
    del st, res
for it in range(2):
  r = d = None
  torch.cuda.synchronize(); t0 = time.perf_counter()
  d = host.to(dev, non_blocking=True); torch.cuda.synchronize(); t1 = time.perf_counter()
  r = ingest.ingest_shard(d, cfg); torch.cuda.synchronize(); t2 = time.perf_counter()
  print(f"single shot: copy {1e3 * (t1 - t0):.0f} ms + ingest {1e3 * (t2 - t1):.0f} ms = {tot / (t2 - t0) / 2**30:.2f} GiB/s incl. copy, {tot / (t2 - t1) / 2**30:.2f} GiB/s resident")
