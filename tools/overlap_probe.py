"""Does the VALU-bound MinHash kernel overlap with the latency-bound DEFLATE match kernels?  Times L4a and L1(FULL only)
back to back on one stream and concurrently on two streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hmse_amd import IngestConfig, corpus, ops

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda:0")
cfg = IngestConfig()
d = torch.from_numpy(corpus.wiki_synth(mib << 20, seed=42)).to(dev)
cuts = ops.l2_cdc(d, cfg)
fo, _ = ops.l3_dedup(ops.l3_sha256(d, cuts))
uniq = (fo == torch.arange(fo.numel(), device=dev)).nonzero().flatten()
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()

def run(concurrent):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if concurrent:
        with torch.cuda.stream(sA):
            sig = ops.l4_minhash(d, cuts, cfg, uniq)
        with torch.cuda.stream(sB):
            out = ops.l1_deflate(d, cuts, cfg, uniq, None)
    else:
        sig = ops.l4_minhash(d, cuts, cfg, uniq)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        out = ops.l1_deflate(d, cuts, cfg, uniq, None)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t2 - t0) * 1e3, (0 if concurrent else (t1 - t0) * 1e3)

for it in range(3):
    s, m = run(False)
    c, _ = run(True)
    print(f"iter {it}: sequential {s:.1f} ms (minhash {m:.1f} + deflate-full {s - m:.1f})   concurrent {c:.1f} ms", flush=True)
