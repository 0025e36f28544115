"""STUDY (test infrastructure): lane-trips of the matcher's state machine per chunk position under filter variants (exact walks)."""
import ctypes as C, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import oracle as O
import lz_study as S
from multiprocessing import Pool
V = {"r3 kernel (K8: 4 hash bits + nib b4, 2/trip)": (0, 2), "K8 as is, 4/trip": (0, 4), "K16: hash4 + nib b4,b5,b6, 1/trip": (1, 1), "K16, 2/trip": (1, 2), "K16, 4/trip": (1, 4), "K16, 8/trip": (1, 8),
     "K8': nib b4,b5, 2/trip": (2, 2), "K8', 4/trip": (2, 4), "K10: K8 + 2 bits of b5, 2/trip": (3, 2), "K10, 4/trip": (3, 4),
     "K10h: 6 hash bits + nib b4, 2/trip": (4, 2), "K10b: 4 hash bits + 6 bits of b4, 2/trip": (5, 2), "K12: hash4 + nib b4,b5, 2/trip": (6, 2), "K12, 4/trip": (6, 4)}
def work(buf):
    S.build(); L = C.CDLL(S.SO); L.study_trips.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
    d = np.frombuffer(buf, np.uint8); cuts = O.cdc(d, O.default_cfg())
    res = {k: np.zeros(8, np.uint64) for k in V}; pos = 0
    for i in range(len(cuts) - 1):
        ch = d[int(cuts[i]):int(cuts[i + 1])]; pos += ch.size
        for k, (fm, g) in V.items(): L.study_trips(ch.ctypes.data, ch.size, 32, fm, g, None, res[k].ctypes.data)
    return pos, res
if __name__ == "__main__":
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    S.build()
    for name, blob in S.sets(mib):
        segs = [blob[i:i + (2 << 20)] for i in range(0, len(blob) - (2 << 20) + 1, 2 << 20)]
        with Pool(8) as p: r = p.map(work, segs)
        pos = sum(x[0] for x in r)
        print(f"== {name}")
        for k in V:
            s = sum(x[1][k].astype(np.int64) for x in r)
            print(f"  {k:48s} probe trips/pos {s[0] / pos:5.2f}  extend trips/pos {s[1] / pos:4.2f}  cands/pos {s[2] / pos:5.2f}  filter-consumed {100 * s[3] / max(1, s[2]):4.1f}%  window probes/pos {s[4] / pos:4.2f}")
