"""Diagnostic: BASELINE configs[1] (FastCDC + SHA-256 dedupe over 1 GB) alone, for a kernel trace:
rocprofv3 --kernel-trace --stats -d gpurun_out/cdc_kt -- python3 tools/cdc_step_trace.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hmse_amd import ABLATIONS, IngestConfig, corpus, ingest, ops
dev = torch.device("cuda:0")
n1 = int(1e9) // (4 << 20) * (4 << 20)
d1 = torch.from_numpy(corpus.wiki_synth(n1, seed=42)).to(dev)
cfg1 = IngestConfig(layers=ABLATIONS["cdc_dedupe"])
so1 = ops.segment_offsets(n1, cfg1.seg_size, dev)
for _ in range(3):
    r1 = ingest.ingest_shard(d1, cfg1, so1, want_stats=False)
torch.cuda.synchronize()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
t0 = time.perf_counter()
for _ in range(steps):
    r1 = None
    r1 = ingest.ingest_shard(d1, cfg1, so1, want_stats=False)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"configs[1]: {dt * 1e3:.3f} ms per 1 GB step = {n1 / dt / 2**30:.1f} GiB/s")
