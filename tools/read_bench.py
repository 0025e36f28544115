"""Time the GPU read path (inflate, assemble, SHA-256 verify) on a wiki-synth shard: python tools/read_bench.py [MiB]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hmse_amd import IngestConfig, _lib, corpus, ingest, ops, read

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
data = torch.from_numpy(corpus.wiki_synth(mib << 20, seed=42)).to(dev)
res = ingest.ingest_shard(data, IngestConfig())
lib = _lib.hip_lib()
lib.hmse_profile_enable(1)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    back = read.reconstruct_shard(res, verify=True)
    torch.cuda.synchronize(); t1 = time.time()
    out = {}
    import ctypes as C
    for name, slot in (("inflate", 16), ("assemble", 17), ("sha256", 3)):
        ms, n = C.c_double(), C.c_uint64()
        lib.hmse_profile_read(slot, C.byref(ms), C.byref(n), 1)
        out[name] = round(ms.value, 2)
    print(f"iter {it}: read path {1e3 * (t1 - t0):.1f} ms = {data.numel() / 2**30 / (t1 - t0):.2f} GiB/s  kernels(ms) {out}  "
          f"stored {res.streams.numel() / 1e6:.1f} MB -> {data.numel() / 1e6:.1f} MB", flush=True)
assert torch.equal(back, data)
print("identical")
