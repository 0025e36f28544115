"""Time one MinHash kernel variant (HMSE_MH_VARIANT) on wiki-synth and print a checksum of the signatures."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hmse_amd import IngestConfig, _lib, corpus, ops
_lib.HIP_LIB_PATH = _lib.HIP_LIB_PATH.replace("libhmse_hip.so", "libhmse_hip_diag.so")  # make -C hmse_amd/csrc libhmse_hip_diag.so
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda:0")
cfg = IngestConfig()
d = torch.from_numpy(corpus.wiki_synth(mib << 20, seed=42)).to(dev)
cuts = ops.l2_cdc(d, cfg)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sig = ops.l4_minhash(d, cuts, cfg)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("variant", os.environ.get("HMSE_MH_VARIANT", "default"), f"{dt * 1e3:.2f} ms  {d.numel() / dt / 1e9:.1f} GB/s",
      hashlib.sha256(sig.cpu().numpy().tobytes()).hexdigest()[:16], flush=True)
