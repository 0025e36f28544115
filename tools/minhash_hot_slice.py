"""STUDY (CPU, uses the oracle's chunker): what share of a chunk's DISTINCT 4-byte shingles (the MinHash kernel looks
each up once in its memo table: one lane-divergent 8-byte gather) lies in the K shingles that occur in most chunks?  If a hot slice
of the memo table kept in LDS covered most look-ups, the gathers — what bounds the kernel — would become LDS probes.
    python tools/minhash_hot_slice.py [MiB]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O
from hmse_amd import corpus

def study(name, data):
    cuts = O.cdc(data, O.default_cfg())
    per = []
    for i in range(len(cuts) - 1):
        c = data[int(cuts[i]):int(cuts[i + 1])]
        if c.size < 4: continue
        x = c[:-3].astype(np.uint32) | (c[1:-2].astype(np.uint32) << 8) | (c[2:-1].astype(np.uint32) << 16) | (c[3:].astype(np.uint32) << 24)
        per.append(np.unique(x))
    allx = np.concatenate(per)
    u, cnt = np.unique(allx, return_counts=True)          # cnt = number of chunks a shingle occurs in
    order = np.argsort(-cnt)
    tot = allx.size
    print(f"{name}: {len(per)} chunks, {tot / len(per):.0f} distinct shingles per chunk, {u.size} distinct shingles in all")
    # a first-half / second-half split: the hot set is learned on the first half of the chunks and used on the second (what a running kernel can do)
    half = len(per) // 2
    u1, c1 = np.unique(np.concatenate(per[:half]), return_counts=True)
    o1 = u1[np.argsort(-c1)]
    second = np.concatenate(per[half:])
    for K in (1024, 2048, 4096, 8192, 16384):
        hot = np.sort(u[order[:K]])
        cov = np.isin(allx, hot).sum() / tot
        hot1 = np.sort(o1[:K])
        cov1 = np.isin(second, hot1).sum() / second.size
        print(f"   hot slice of {K:6d} shingles ({K * 8 >> 10:3d} KiB of LDS): covers {100 * cov:5.1f} % of the look-ups (oracle choice); {100 * cov1:5.1f} % when learned on the first half of the chunks")

if __name__ == "__main__":
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    for prof in ("wikipedia", "code"):
        study("wiki-synth " + prof, corpus.load(prof, mib << 20, seed=42)[0])
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import real_bytes
    study("real bytes (tests/real_bytes.py)", real_bytes.gather(mib << 20))
