"""Run by tests/test_gpu_parity.py in a child process with HMSE_ENC_FORCE_SPILL=1 (the library reads the variable once): every FULL
record then takes the encode kernel's spill path — finished windows of the bit image go to the global scratch and move into the
record's slot when the last token has been read — and every stream must still equal the oracle's."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import oracle as orc   # noqa: E402  (test infrastructure)
from hmse_amd import IngestConfig, ops   # noqa: E402
from conftest import words_text   # noqa: E402
from test_gpu_parity import ocfg   # noqa: E402

assert os.environ.get("HMSE_ENC_FORCE_SPILL") == "1"
dev = torch.device("cuda:0")
cfg = IngestConfig()
rng = np.random.Generator(np.random.PCG64(99))
text = words_text(60000, seed=9)
parts = []
for L, hi in ((12288, 64), (12288, 128), (9000, 200), (32768, 128), (30000, 64), (20000, 32), (16384, 250), (8192, 16), (700, 4), (5, 2)):
    parts.append(rng.integers(0, hi, L, dtype=np.uint8))
    parts.append(text[100 * len(parts): 100 * len(parts) + 2000 + 1500 * len(parts)])
for L in list(range(6450, 6650, 16)) + list(range(12950, 13150, 16)):
    parts.append(rng.integers(0, 32, L, dtype=np.uint8))
variant = parts[1].copy(); variant[::97] = 35
parts.append(variant)
data = np.concatenate(parts)
cuts = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.uint64)
base = np.full(len(parts), -1, dtype=np.int64)
base[len(parts) - 1] = 1          # one DELTA record beside the FULL ones
want_out, want_off, want_kind = orc.deflate_chunks(data, cuts, ocfg(orc, cfg), None, base)
for _ in range(2):
    out, off, kind = ops.l1_deflate(torch.from_numpy(data).to(dev), torch.from_numpy(cuts.astype(np.int64)).to(dev), cfg, None, torch.from_numpy(base).to(dev))
    assert np.array_equal(off.cpu().numpy().astype(np.uint64), want_off)
    assert np.array_equal(kind.cpu().numpy(), want_kind)
    assert np.array_equal(out.cpu().numpy(), want_out)
sizes = np.diff(want_off.astype(np.int64))
assert (sizes > 8192).sum() >= 4
print(f"spill path OK: {len(parts)} records, {int((sizes > 4096).sum())} of them longer than one window")
