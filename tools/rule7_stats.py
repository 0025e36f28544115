"""STUDY (test infrastructure): rule 7 keeps a delta of at most a fifth of the chunk WITHOUT computing the chunk's FULL encoding
(README.md:1328, 2175).  How many of those quick accepts would have lost the round-1/2 test `delta + 8 < full`, and what do they cost?
    python tools/rule7_stats.py [MiB per profile]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from multiprocessing import Pool
from oracle import oracle as O
from hmse_amd import corpus

def mh(args):
    data, cuts, ids = args
    return O.minhash_chunks(data, cuts, O.default_cfg(), ids)
def enc(args):
    data, cuts, pairs = args
    cfg = O.default_cfg(); out = []
    for c, b in pairs:
        ch = data[int(cuts[c]):int(cuts[c + 1])]; d = data[int(cuts[b]):int(cuts[b + 1])]
        n2 = len(O.deflate(ch, cfg, d)); n1 = len(O.deflate(ch, cfg))
        out.append((ch.size, n1, n2))
    return out
if __name__ == "__main__":
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 192
    for prof in ("wikipedia", "news", "code"):
        data = corpus.load(prof, mib << 20, seed=42)[0]
        cfg = O.default_cfg(); cuts = O.cdc(data, cfg); fo, _ = O.dedup(O.sha256_chunks(data, cuts))
        uniq = np.nonzero(fo == np.arange(len(fo)))[0].astype(np.uint64)
        with Pool(8) as p:
            sl = np.array_split(uniq, 32)
            sig = np.concatenate(p.map(mh, [(data, cuts, s) for s in sl]))
            _, base = O.lsh(sig, cfg)
            pairs = [(int(uniq[k]), int(uniq[base[k]])) for k in np.nonzero(base >= 0)[0]]
            r = sum(p.map(enc, [(data, cuts, pairs[i::32]) for i in range(32)]), [])
        r = np.array(r, np.int64)
        L, n1, n2 = r[:, 0], r[:, 1], r[:, 2]
        quick = 5 * n2 <= L
        lose = quick & (n2 + 8 >= n1)
        cost = int((n2[lose] + 8 - n1[lose]).sum())
        stored = int(np.where(quick | (n2 + 8 < n1), n2 + 8, n1).sum())
        print(f"{prof}: {len(uniq)} stored chunks, {len(r)} with a base; quick accepts (5 * delta <= chunk) {int(quick.sum())}; of those NOT net-saving "
              f"(delta + 8 >= full) {int(lose.sum())}, costing {cost} bytes = {100 * cost / max(1, stored):.4f} % of these chunks' stored bytes; "
              f"not quick but kept (delta + 8 < full) {int((~quick & (n2 + 8 < n1)).sum())}, refused {int((~quick & (n2 + 8 >= n1)).sum())}")
