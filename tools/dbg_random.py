import numpy as np, torch, sys
from hmse_amd import IngestConfig, ops, ingest
cfg=IngestConfig(); dev=torch.device("cuda:0")
rng=np.random.Generator(np.random.PCG64(7))
for mib in (32, 64, 128, 256):
    d=torch.from_numpy(rng.integers(0,256,mib<<20,dtype=np.uint8)).to(dev)
    cuts=ops.l2_cdc(d,cfg)
    out,off,kind=ops.l1_deflate(d,cuts,cfg)
    ln=(off[1:]-off[:-1]); L=cuts[1:]-cuts[:-1]
    bad=(ln!=L+5).nonzero().flatten()
    print(mib,"MiB chunks",L.numel(),"total",int(off[-1]),"expected",int((L+5).sum()),"bad",bad.numel(), "first bad", bad[:5].tolist(), "their L", L[bad[:5]].tolist(), "len", ln[bad[:5]].tolist())
