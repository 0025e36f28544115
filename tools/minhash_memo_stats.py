"""Hit rate of the MinHash memo table (diagnostic build): python tools/minhash_memo_stats.py [MiB] [profile]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hmse_amd import IngestConfig, _lib, corpus, ingest, ops
DIAG = os.environ.get("MH_DIAG", "1") == "1"   # MH_DIAG=0: the product library (timings only; the counters exist in the diagnostic build)
if os.environ.get("HMSE_LIB"):
    _lib.HIP_LIB_PATH = os.path.join(ROOT, "hmse_amd", "csrc", os.environ["HMSE_LIB"])
elif DIAG:
    _lib.HIP_LIB_PATH = os.path.join(ROOT, "hmse_amd", "csrc", "libhmse_hip_diag.so")
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
prof = sys.argv[2] if len(sys.argv) > 2 else "wikipedia"
dev = torch.device("cuda:0")
cfg = IngestConfig()
n = (mib << 20) // cfg.seg_size * cfg.seg_size
if prof == "prng":     # no repeating 4-grams: the table fills up, wavefronts stop looking after MH_SAMPLE lookups per pass
    data = torch.randint(0, 256, (n,), dtype=torch.uint8, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
else:
    data = torch.from_numpy(corpus.load(prof, n, seed=42)[0]).to(dev)
cuts = ops.l2_cdc(data, cfg)
dg = ops.l3_sha256(data, cuts)
fo, _ = ops.l3_dedup(dg)
uniq = (fo == torch.arange(cuts.numel() - 1, device=dev)).nonzero().flatten()
nb = ops.workspace_bytes(ops.STAGE_MINHASH, uniq.numel(), cfg)
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
sig = torch.empty((uniq.numel(), 128), dtype=torch.int32, device=dev)
c = cfg.to_c()
lib = _lib.hip_lib()
for memo in (True, False, True, False):
    best = 1e9
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lib.hmse_l4_minhash(C.c_void_p(data.data_ptr()), data.numel(), C.c_void_p(cuts.data_ptr()), C.c_void_p(uniq.data_ptr()), uniq.numel(), C.byref(c),
                            C.c_void_p(sig.data_ptr()), C.c_void_p(ws.data_ptr()), nb if memo else 256, None)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"memo {memo}: {best * 1e3:8.2f} ms for {uniq.numel()} chunks", flush=True)
    if memo and DIAG and int(os.environ.get('HMSE_MH_PROBE', '0')) & 4:
        cnt = ws[:32].view(torch.int32).tolist()
        print(f"  entries {cnt[0]}  lookups {cnt[1] & 0xFFFFFFFF}  todo after lookups {cnt[2] & 0xFFFFFFFF} ({(cnt[2] & 0xFFFFFFFF) / max(1, cnt[1] & 0xFFFFFFFF):.4f})  distinct in looked-up parts {cnt[4] & 0xFFFFFFFF}  "
              f"passes {cnt[5]}  unresolved seeds {cnt[3]} ({cnt[3] / max(1, cnt[5]):.2f} per pass)")
