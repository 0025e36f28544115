"""Timing of the full pipeline on adversarial 256 MiB inputs (nothing may take seconds)."""
import time
import numpy as np
import torch
from hmse_amd import IngestConfig, ingest

dev = torch.device("cuda:0")
cfg = IngestConfig()
n = 256 << 20
rng = np.random.Generator(np.random.PCG64(7))
cases = {
    "zeros": lambda: np.zeros(n, np.uint8),
    "binary-random": lambda: rng.integers(0, 2, n, dtype=np.uint8),
    "random": lambda: rng.integers(0, 256, n, dtype=np.uint8),
    "period-97+noise": lambda: (np.tile(rng.integers(97, 123, 97, dtype=np.uint8), n // 97 + 1)[:n] ^ (rng.random(n) < 0.001).astype(np.uint8)),
    "long-runs": lambda: np.repeat(rng.integers(0, 256, n // 1000 + 1, dtype=np.uint8), 1000)[:n],
    "4-symbol": lambda: rng.integers(0, 4, n, dtype=np.uint8) + 65,
}
for name, gen in cases.items():
    d = torch.from_numpy(np.ascontiguousarray(gen())).to(dev)
    torch.cuda.synchronize(); t0 = time.time()
    r = ingest.ingest_shard(d, cfg)
    torch.cuda.synchronize(); dt = time.time() - t0
    st = ingest.merge_stats([r.stats])
    print(f"{name:18s} {dt*1e3:9.1f} ms  chunks {st['chunks']:7d} unique {st['unique']:7d} delta {st['delta']:6d} cf {st['cf']:.2f}", flush=True)
