"""Diagnostic: per-phase clock shares of the MinHash kernel (thread 0 of every workgroup; uses libhmse_hip_diag.so).
python tools/minhash_stamps.py [MiB]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hmse_amd import IngestConfig, _lib, corpus, ops
_lib.HIP_LIB_PATH = _lib.HIP_LIB_PATH.replace("libhmse_hip.so", "libhmse_hip_diag.so")   # make -C hmse_amd/csrc libhmse_hip_diag.so
lib = _lib.hip_lib()
lib.hmse_debug_minhash_stamps.argtypes = [C.c_void_p, C.c_int]
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
cfg = IngestConfig()
d = torch.from_numpy(corpus.wiki_synth(mib << 20, seed=42)).to(dev)
cuts = ops.l2_cdc(d, cfg)
fo, _ = ops.l3_dedup(ops.l3_sha256(d, cuts))
uniq = (fo == torch.arange(fo.numel(), device=dev)).nonzero().flatten()
ops.l4_minhash(d, cuts, cfg, uniq); torch.cuda.synchronize()
buf = np.zeros(16, dtype=np.uint64)
lib.hmse_debug_minhash_stamps(buf.ctypes.data, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.l4_minhash(d, cuts, cfg, uniq); e1.record(); torch.cuda.synchronize()
lib.hmse_debug_minhash_stamps(buf.ctypes.data, 0)
names = ["prologue: chunk metadata", "clear + room-in-table query + barrier", "set inserts + barrier", "compaction", "memo look-ups", "tail (misses)", "barrier behind the tail",
         "combine + re-evaluation", "signature out"]
n = int(buf[15]); tot = float(buf[:9].sum())
print(f"{mib} MiB: {n} chunks, kernel {e0.elapsed_time(e1):.2f} ms, {tot / n:.0f} clocks per chunk (thread 0)")
for i, nm in enumerate(names):
    print(f"  {nm:42s} {100 * float(buf[i]) / tot:5.1f} %   {float(buf[i]) / n:8.0f} clk")
