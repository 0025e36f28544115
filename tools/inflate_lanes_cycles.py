"""Where the lane-per-stream inflate kernel spends its cycles (diagnostic build: make -C hmse_amd/csrc libhmse_hip_diag.so):
python tools/inflate_lanes_cycles.py [MiB]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hmse_amd import IngestConfig, _lib, corpus, ingest, ops
_lib.HIP_LIB_PATH = os.path.join(ROOT, "hmse_amd", "csrc", "libhmse_hip_diag.so")
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda:0")
cfg = IngestConfig()
n = (mib << 20) // cfg.seg_size * cfg.seg_size
data = torch.from_numpy(corpus.load("wikipedia", n, seed=42)[0]).to(dev)
res = ingest.ingest_shard(data, cfg)
lens = (res.cuts[1:] - res.cuts[:-1])[res.uniq_ids]
trace = torch.zeros(32, dtype=torch.int64).pin_memory()
lib = _lib.hip_lib()
lib.hmsedbg_inflate_trace(C.c_void_p(trace.data_ptr()))
ops.l1_inflate_mode(2)
ops.l1_inflate(res.streams, res.stream_off, res.kind, res.base, lens)
torch.cuda.synchronize()
trace.zero_()
ops.l1_inflate(res.streams, res.stream_off, res.kind, res.base, lens)
torch.cuda.synchronize()
t = trace.tolist()
waves = min((res.kind.numel() + 63) // 64, 2048)
names = ["trips", "header rounds", "publish", "header", "poll", "decode", "memory cluster", "lane-trips decoding", "lane-trips copying", "lane-trips waiting for header", "lane-trips out of work", "lane-trips blocked on base", "header: pull", "header: block header + staging", "header: code lengths", "header: tables + placement"]
print(f"{res.kind.numel()} records, {waves} wavefronts")
for i, nm in enumerate(names):
    print(f"  {nm:30s} {t[i]:16d}   per wavefront {t[i] / waves:14.0f}" + (f"   per trip {t[i] / max(t[0], 1):8.1f}" if i >= 2 else ""))
print(f"  lanes decoding per trip: {t[7] / max(t[0], 1):.1f} of 64;  cycle counter ticks per trip: {sum(t[2:7]) / max(t[0], 1):.0f}")
