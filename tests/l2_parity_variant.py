import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hmse_amd import IngestConfig, _lib, corpus, ops
v=sys.argv[1]
if v!="base": _lib.HIP_LIB_PATH=_lib.HIP_LIB_PATH.replace("libhmse_hip.so", f"libhmse_hip_l2_{v}.so")
from oracle import oracle as O
from dataclasses import asdict
cfg=IngestConfig(); dev=torch.device("cuda:0")
for n in (0,1,63,2047,32768,32769,1000003, 5*(1<<20)+12345):
    d=corpus.wiki_synth(max(n,1<<20))[:n].copy()
    want=O.cdc(d,O.default_cfg(**asdict(cfg)))
    got=ops.l2_cdc(torch.from_numpy(d).to(dev),cfg).cpu().numpy().astype(np.uint64)
    assert np.array_equal(got,want),(v,n)
print(v,"parity ok")
