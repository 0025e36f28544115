"""One-off parity run at a size the CPU oracle needs minutes for: every output of ingest_shard against the oracle pipeline.
python tests/parity_at_scale.py [MiB] [profile]   (the committed tests do the same at <= 5 MiB)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from hmse_amd import IngestConfig, corpus, ingest
from oracle import oracle as O
from test_gpu_ingest import oracle_pipeline

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 48
prof = sys.argv[2] if len(sys.argv) > 2 else "wikipedia"
O.build()
cfg = IngestConfig()
data, src = (corpus.random_bytes(mib << 20), "PRNG bytes (seed 0xDEADBEEF, VALIDATION_METHODS.md:213)") if prof == "random" else corpus.load(prof, mib << 20, seed=42)
t0 = time.time()
res = ingest.ingest_shard(torch.from_numpy(data).to("cuda:0"), cfg)
torch.cuda.synchronize()
print(f"GPU ingest of {mib} MiB {src}: {time.time() - t0:.2f} s (cold)", flush=True)
t0 = time.time()
_, (o,) = oracle_pipeline(O, data, cfg)
print(f"oracle pipeline: {time.time() - t0:.1f} s on one core", flush=True)
checks = [("cuts", res.cuts.cpu().numpy().astype(np.uint64), o["cuts"]), ("digests", res.digests.cpu().numpy(), o["dg"]),
          ("first_occ", res.first_occ.cpu().numpy().astype(np.uint64), o["fo"]), ("uniq_ids", res.uniq_ids.cpu().numpy().astype(np.uint64), o["uniq"]),
          ("signatures", res.sig.cpu().numpy().view(np.uint32), o["sig"]), ("bases", res.base.cpu().numpy(), o["base"]),
          ("kinds", res.kind.cpu().numpy(), o["kind"]), ("stream offsets", res.stream_off.cpu().numpy().astype(np.uint64), o["off"]),
          ("streams", res.streams.cpu().numpy(), o["out"])]
bad = 0
for name, got, want in checks:
    ok = got.shape == want.shape and np.array_equal(got, want)
    bad += not ok
    print(f"  {name:15s} {'bit-exact' if ok else 'MISMATCH'}  ({want.size} elements)", flush=True)
st = res.stats
print(f"chunks {st['chunks']}  unique {st['unique']}  delta {st['delta']}  pointer {st['pointer']}  stored {st['stored_bytes']} B")
sys.exit(1 if bad else 0)
