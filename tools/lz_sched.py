"""STUDY (test infrastructure): scheduling model of the plain-job matcher — wave-trips per job under hand-out variants."""
import ctypes as C, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import oracle as O
import lz_study as S
from multiprocessing import Pool
V = {"r3: batch 16": (16, 0, 0), "no first-of-bucket ranks in the queue": (16, 48, 95 << 8), "3 classes, no kmax 0": (16, 48, 96 << 8), "4 classes, no kmax 0": (16, 48, 97 << 8), "8 classes, no kmax 0": (16, 48, 98 << 8), "8 classes, no kmax 0, batch 8": (8, 48, 98 << 8), "8 classes, no kmax 0, batch 24": (24, 48, 98 << 8), "33 classes by chain length": (16, 48, 99 << 8), "8 classes by chain length": (16, 48, 98 << 8), "8 classes + prefetch": (16, 49, 98 << 8), "no extend trips at all": (16 + 256, 0, 0), "extend 64 B per trip": (16 + 512, 0, 0), "no extend trips + perfect LF": (16 + 256, 16, 0), "perfect longest-first": (16, 16, 0), "chains >= 8 first": (16, 32, 8 << 8), "chains >= 16 first": (16, 32, 16 << 8), "chains >= 24 first": (16, 32, 24 << 8), "four classes by chain length": (16, 48, 0),
     "buckets >= 8 first, reversed": (16, 64, 8 << 8), "buckets >= 16 first, reversed": (16, 64, 16 << 8), "buckets >= 4 first, reversed": (16, 64, 4 << 8), "four bucket classes, reversed": (16, 80, 0),
     "perfect LF + prefetch slot": (16, 17, 0), "four classes + prefetch": (16, 49, 0), "batch 1": (1, 0, 0), "batch 8": (8, 0, 0), "prefetch slot, batch 16": (16, 1, 0), "prefetch slot, batch 32": (32, 1, 0), "coop tail <= 8": (16, 2, 8), "coop tail <= 16": (16, 2, 16),
     "prefetch + coop tail <= 8": (16, 3, 8), "prefetch + coop tail <= 16": (16, 3, 16)}
def work(buf):
    S.build(); L = C.CDLL(S.SO); L.study_sched.argtypes = [C.c_void_p, C.c_uint32] + [C.c_uint32] * 5 + [C.c_void_p]
    d = np.frombuffer(buf, np.uint8); cuts = O.cdc(d, O.default_cfg())
    res = {k: np.zeros(8, np.uint64) for k in V}; pos = 0; nj = 0
    for i in range(len(cuts) - 1):
        ch = d[int(cuts[i]):int(cuts[i + 1])]
        if ch.size > 10048: continue          # class S
        pos += ch.size; nj += 1
        for k, (b, var, ct) in V.items(): L.study_sched(ch.ctypes.data, ch.size, 32, 16, b, var, ct, res[k].ctypes.data)
    return pos, nj, res
if __name__ == "__main__":
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    S.build()
    for name, blob in S.sets(mib):
        if not name.startswith("synth_w") and name != "py": continue
        segs = [blob[i:i + (2 << 20)] for i in range(0, len(blob) - (2 << 20) + 1, 2 << 20)]
        with Pool(8) as p: r = p.map(work, segs)
        pos = sum(x[0] for x in r); nj = sum(x[1] for x in r)
        print(f"== {name}: {nj} class-S jobs, {pos / nj:.0f} positions per job")
        for k in V:
            s = sum(x[2][k].astype(np.int64) for x in r)
            lt = s[2] + s[3] + s[4]
            print(f"  {k:30s} wave-trips per job and wavefront {s[0] / nj / 16:6.1f}  pulls {s[1] / nj / 16:5.1f}  lanes walking {100 * s[2] / lt:4.1f}%  idle {100 * s[3] / lt:4.1f}%  done {100 * s[4] / lt:4.1f}%")
