"""Diagnostic: per-phase clock shares of the DEFLATE kernel (uses the -DHMSE_DFL_STAMPS build)."""
import ctypes as C
import sys

import numpy as np
import torch

from hmse_amd import IngestConfig, _lib, corpus, ops

import os
_lib.HIP_LIB_PATH = _lib.HIP_LIB_PATH.replace("libhmse_hip.so", os.environ.get("HMSE_STAMPS_LIB", "libhmse_hip_stamps.so"))
lib = _lib.hip_lib()
lib.hmse_debug_deflate_stamps.argtypes = [C.c_void_p, C.c_int]
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = IngestConfig()
dev = torch.device("cuda:0")
d = torch.from_numpy(corpus.wiki_synth(mib << 20)).to(dev)
cuts = ops.l2_cdc(d, cfg)
dg = ops.l3_sha256(d, cuts)
fo, _ = ops.l3_dedup(dg)
uniq = (fo == torch.arange(fo.numel(), device=dev)).nonzero().flatten()
sig = ops.l4_minhash(d, cuts, cfg, uniq)
_, base = ops.l4_lsh(sig, cfg)
ops.l1_deflate(d, cuts, cfg, uniq, base)
torch.cuda.synchronize()
buf = np.zeros(144, dtype=np.uint64)
lib.hmse_debug_deflate_stamps(buf.ctypes.data, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.l1_deflate(d, cuts, cfg, uniq, base); e1.record(); torch.cuda.synchronize()
lib.hmse_debug_deflate_stamps(buf.ctypes.data, 0)
ebuf = np.zeros(8, dtype=np.uint64)
lib.hmse_debug_encode_stamps.argtypes = [C.c_void_p, C.c_int]
lib.hmse_debug_encode_stamps(ebuf.ctypes.data, 0)
names = ["0 load+clear", "1 hist+scan", "2 scatter+rank", "3 match: state machine", "4 parse: stitch", "5 tokens + histograms", "6 match: anchors + hinted positions", "7 -", "8 hand-out order lists", "9 record out", "10 job fetch", "11 parse: next-pointers", "12 parse: speculative walks"]
lens = (cuts[1:] - cuts[:-1])[uniq]
hb = base >= 0
Tfull = lens
Tdelta = (lens + torch.clamp(lens[base.clamp(min=0)], max=32768))[hb]
allT = torch.cat([Tfull, Tdelta]).cpu().numpy()
allL = torch.cat([lens, lens[hb]]).cpu().numpy()
import numpy as _np
_c = ops.DEFLATE_CLASS_CAPS
cls_ = _np.where(allT <= _c[0], 0, _np.where(allT <= _c[1], 1, _np.where(allT <= _c[2], 2, _np.where(allT <= _c[3], 3, _np.where(allT <= _c[4], 4, 5)))))
for ci_, nm in enumerate(("S", "S2", "SG", "SG2", "SG3", "B")):
    m_ = cls_ == ci_
    print("class %-6s jobs %6d  window bytes %6.1f MB  chunk bytes %6.1f MB" % (nm, m_.sum(), allT[m_].sum() / 1e6, allL[m_].sum() / 1e6))
print("deflate op ms", e0.elapsed_time(e1), "jobs", uniq.numel(), "+", int((base >= 0).sum()))
for c, cn in enumerate([f"plain S (T <= {_c[0]})", f"plain S2, SG (T <= {_c[2]})", "plain SG2, SG3, B", "DICT S", "DICT S2, SG", "DICT SG2, SG3, B"]):
    row = buf[c * 24:(c + 1) * 24].astype(np.float64)
    occ = row[16:22].copy(); row = row[:16]
    trips, positions, jobs = row[13], row[14], row[15]
    walked, usable = row[7], row[8]
    row[7:9] = 0
    row[13:] = 0
    if walked:
        print(cn, "positions walked by lane 0: %d, %.1f %% with a usable partial hint" % (walked, 100 * usable / walked))
    tot = row.sum()
    if tot == 0:
        continue
    print(cn, "jobs %d  chunk positions %.1f M  trips of wavefront 0 through the matcher: %.1f per job = %.2f per 64 positions" % (jobs, positions / 1e6, trips / max(jobs, 1), 64 * trips / max(positions, 1)))
    print(cn, "total Mclk %.1f" % (tot / 1e6))
    if trips:
        print(cn, "matcher blocks per trip of wavefront 0: FETCH runs in %.0f %% of the trips with %.1f lanes, PROBE %.0f %% with %.1f lanes, EXTEND %.0f %% with %.1f lanes"
              % (100 * occ[0] / trips, occ[1] / max(occ[0], 1), 100 * occ[2] / trips, occ[3] / max(occ[2], 1), 100 * occ[4] / trips, occ[5] / max(occ[4], 1)))
    for i, nm in enumerate(names):
        print("   %-18s %6.2f %%" % (nm, 100 * row[i] / tot))

et = ebuf[:6].astype(np.float64)
if et.sum():
    print("encode kernel: %d records (both launches, warm-up included), thread 0's clocks per phase:" % int(ebuf[7]))
    for nm, v in zip(("load histograms", "trees (lit/len, dist, fixed cost)", "rle + code-length tree + decide", "code tables + image clear", "emit", "copy-out"), et):
        print("   %-34s %6.2f %%   %8.0f clk per record" % (nm, 100 * v / et.sum(), v / max(1, int(ebuf[7]))))
