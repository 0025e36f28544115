"""A/B of the two hmse_l1_inflate decoders on one ingested shard: python tools/inflate_ab.py [MiB] [profile]
Prints ms per decoder (best of 3) for the inflate call alone, checks the outputs are identical and that the whole shard
reassembles to the input."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hmse_amd import IngestConfig, _lib, corpus, ingest, ops, read
if os.environ.get("HMSE_LIB"):   # an experimental build of the library (same ABI)
    _lib.HIP_LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hmse_amd", "csrc", os.environ["HMSE_LIB"])

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
prof = sys.argv[2] if len(sys.argv) > 2 else "wikipedia"
dev = torch.device("cuda:0")
cfg = IngestConfig()
n = (mib << 20) // cfg.seg_size * cfg.seg_size
data = torch.from_numpy(corpus.load(prof, n, seed=42)[0]).to(dev)
res = ingest.ingest_shard(data, cfg)
lens = (res.cuts[1:] - res.cuts[:-1])[res.uniq_ids]
print(f"{n / 2**30:.2f} GiB {prof}: {res.kind.numel()} stored records, {int((res.kind == 2).sum())} DELTA, {res.streams.numel() / 2**20:.0f} MiB of streams", flush=True)
outs = {}
for mode in (1, 2, 1, 2):
    ops.l1_inflate_mode(mode)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        raw, raw_off, ok = ops.l1_inflate(res.streams, res.stream_off, res.kind, res.base, lens)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"mode {mode}: {best * 1e3:8.2f} ms  = {raw.numel() / best / 2**30:6.1f} GiB/s of raw output", flush=True)
    if mode in outs:
        assert torch.equal(outs[mode], raw)
    outs[mode] = raw
    del raw
assert torch.equal(outs[1], outs[2]), "decoders disagree"
del outs
ops.l1_inflate_mode(2)
back = read.reconstruct_shard(res, verify=True)
assert torch.equal(back, data)
print("lane-per-stream decoder: shard reassembled, SHA-256 of every chunk verified, identical to the input")
ops.l1_inflate_mode(0)
