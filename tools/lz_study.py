"""STUDY (test infrastructure: loads oracle/): stored bytes and matcher work of encoder-definition variants, per 8 KiB FastCDC
chunk, FULL records, on real text found in the image and on the synthetic corpus profiles.  VERDICT r3 item 1.
    python tools/lz_study.py [MiB per set]"""
import ctypes as C, os, subprocess, sys, zlib, time
import numpy as np
from multiprocessing import Pool
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O
from depth_real_text import collect

SO = os.path.join(ROOT, "tools", "study", "_build", "libstudy.so")
def build():
    src = os.path.join(ROOT, "tools", "study", "lz_study.c")
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    deps = [src, os.path.join(ROOT, "oracle", "hmse_oracle_deflate.c")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-o", SO, src])
VARIANTS = {  # name -> (D, mode, K1, C1, iters)
    "all_d32": (32, 0, 0, 0, 0),
    "2pass_k1_c258_i1": (32, 1, 1, 258, 1),
    "2pass_k1_c32_i1": (32, 1, 1, 32, 1),
    "2pass_k1_c32_i2": (32, 1, 1, 32, 2),
    "2pass_k1_c32_i3": (32, 1, 1, 32, 3),
    "2pass_k1_c32_conv": (32, 1, 1, 32, 0),
    "2pass_k2_c32_i1": (32, 1, 2, 32, 1),
    "2pass_k4_c32_i1": (32, 1, 4, 32, 1),
    "2pass_k4_c32_i2": (32, 1, 4, 32, 2),
}
def work(buf):
    build(); L = C.CDLL(SO); L.study_deflate.restype = C.c_int64; L.study_deflate.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    d = np.frombuffer(buf, np.uint8); cfg = O.default_cfg(); cuts = O.cdc(d, cfg)
    out = np.zeros(70000, np.uint8); st = np.zeros(8, np.uint64)
    res = {k: np.zeros(10, np.int64) for k in VARIANTS}; z = 0
    for i in range(len(cuts) - 1):
        a, b = int(cuts[i]), int(cuts[i + 1]); ch = d[a:b]
        co = zlib.compressobj(9, zlib.DEFLATED, -15, 9); z += len(co.compress(ch.tobytes()) + co.flush())
        for k, v in VARIANTS.items():
            pr = np.array(list(v) + [0, 0, 0], np.uint32)
            n = L.study_deflate(ch.ctypes.data, b - a, None, 0, pr.ctypes.data, st.ctypes.data, out.ctypes.data, 70000)
            assert n > 0
            if k == "all_d32" and i % 16 == 0: assert bytes(out[:n]) == O.deflate(ch, cfg)
            res[k][0] += n; res[k][1:9] += st.astype(np.int64); res[k][9] += 1
    return z, res
def sets(mib):
    yield "py", collect(['/usr/lib/python3/dist-packages', '/usr/local/lib/python3.10/dist-packages'], ('.py',), mib << 20)
    yield "c_headers", collect(['/opt/rocm/include', '/usr/include'], ('.h', '.hpp'), mib << 20)
    yield "docs", collect(['/usr/share', '/usr/local/lib/python3.10/dist-packages', '/opt'], ('.txt', '.md', '.rst', '.html', '.json', '.xml', '.yaml'), mib << 20)
    from hmse_amd import corpus
    for prof in ("wikipedia", "code"):
        yield "synth_" + prof, corpus.load(prof, mib << 20, seed=42)[0].tobytes()
if __name__ == "__main__":
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    build()
    for name, blob in sets(mib):
        segs = [blob[i:i + (2 << 20)] for i in range(0, len(blob) - (2 << 20) + 1, 2 << 20)]
        t = time.time()
        with Pool(8) as p: r = p.map(work, segs)
        z = sum(x[0] for x in r)
        print(f"== {name}: {len(segs) * 2} MiB, zlib-9 per-chunk CF {len(segs) * (2 << 20) / z:.3f} ({time.time() - t:.0f}s)")
        base = None
        for k in VARIANTS:
            s = sum(x[1][k] for x in r); n, pos, wk, cf, c1, tok, it, rd, rdu, nch = [int(v) for v in s]
            if base is None: base = (n, cf)
            print(f"  {k:22s} bytes vs zlib9 {100 * (n - z) / z:+.3f}%  vs all_d32 {100 * (n - base[0]) / base[0]:+.3f}%  walked {100 * wk / pos:5.1f}%  "
                  f"full-walk cands/pos {cf / pos:5.2f} (x{cf / base[1]:.2f})  pass1 cands/pos {c1 / pos:4.2f}  read {100 * rd / pos:4.1f}%  read-unwalked {100 * rdu / pos:4.2f}%  iters/chunk {it / nch:.2f}", flush=True)
