import sys
import torch
from hmse_amd import IngestConfig, _lib, corpus, ops
if len(sys.argv) > 1 and sys.argv[1] != "base":
    _lib.HIP_LIB_PATH = _lib.HIP_LIB_PATH.replace("libhmse_hip.so", f"libhmse_hip_l2_{sys.argv[1]}.so")
cfg = IngestConfig(); dev = torch.device("cuda:0")
d = torch.from_numpy(corpus.wiki_synth(4096 << 20)).to(dev)
lib = _lib.hip_lib(); lib.hmse_profile_enable(1)
import ctypes as C
for _ in range(6): cuts = ops.l2_cdc(d, cfg)
torch.cuda.synchronize()
ms, n = C.c_double(), C.c_uint64(); lib.hmse_profile_read(2, C.byref(ms), C.byref(n), 1)
print(sys.argv[1] if len(sys.argv) > 1 else "base", "hash kernel avg ms", ms.value / n.value, "GB/s", d.numel() / (ms.value / n.value * 1e-3) / 1e9, "chunks", cuts.numel() - 1)
