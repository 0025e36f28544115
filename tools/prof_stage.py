"""Run one stage op a few times (for rocprofv3 --pmc / --kernel-trace runs)."""
import sys
import torch
from hmse_amd import IngestConfig, corpus, ops
mib = int(sys.argv[1]); stage = sys.argv[2]
cfg = IngestConfig(); dev = torch.device("cuda:0")
d = torch.from_numpy(corpus.wiki_synth(mib << 20)).to(dev)
cuts = ops.l2_cdc(d, cfg)
if stage == "l2":
    for _ in range(3): ops.l2_cdc(d, cfg)
elif stage == "l3":
    for _ in range(3): ops.l3_sha256(d, cuts)
else:
    dg = ops.l3_sha256(d, cuts); fo, _ = ops.l3_dedup(dg)
    uniq = (fo == torch.arange(fo.numel(), device=dev)).nonzero().flatten()
    if stage == "l4":
        for _ in range(3): ops.l4_minhash(d, cuts, cfg, uniq)
    else:
        sig = ops.l4_minhash(d, cuts, cfg, uniq); _, base = ops.l4_lsh(sig, cfg)
        for _ in range(2): ops.l1_deflate(d, cuts, cfg, uniq, base)
torch.cuda.synchronize()
