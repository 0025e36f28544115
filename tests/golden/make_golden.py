"""Regenerates tests/golden/*.json from the CPU oracle + the host corpus generator.

The reference (1Jamie/HMSE) holds no golden vectors for this path, so these
fixtures pin THIS build's definitions: a change to the Gear table, the masks, the MinHash/LSH
arithmetic, the corpus generator or the DEFLATE encoder definition shows up as a fixture diff.
Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hmse_amd import corpus  # noqa: E402
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def h(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def variants_dataset(src):
    rng = np.random.Generator(np.random.PCG64(42))
    basechunk = np.array(src[100_000:300_000])
    parts = [basechunk]
    for _ in range(5):
        v = basechunk.copy()
        idx = rng.integers(0, v.size, v.size // 200)
        v[idx] = rng.integers(97, 123, idx.size, dtype=np.uint8)
        parts.append(v)
    parts.append(basechunk)  # one exact duplicate
    return np.concatenate(parts)


def main():
    out = {}
    g = O.gear_table()
    out["gear_table"] = {"sha256": h(g), "first4": [int(x) for x in g[:4]], "last": int(g[255])}
    for name, cfg in (("default", O.default_cfg()), ("reference_1_4_16", O.default_cfg(min_size=1024, avg_size=4096, max_size=16384))):
        ms, ml = O.cdc_masks(cfg)
        out[f"masks_{name}"] = [f"{ms:#018x}", f"{ml:#018x}"]
    data = corpus.wiki_synth(1 << 20, seed=42)
    out["wiki_synth_seed42_first_MiB_sha256"] = h(data)
    data3 = corpus.wiki_synth(3 << 20, seed=42)
    for name, cfg in (("default", O.default_cfg(seg_size=1 << 20)), ("seed1", O.default_cfg(seg_size=1 << 20, seed_base=1)),
                      ("reference_1_4_16", O.default_cfg(min_size=1024, avg_size=4096, max_size=16384, seg_size=1 << 20))):
        cuts = O.cdc(data3, cfg)
        dg = O.sha256_chunks(data3, cuts)
        fo, rc = O.dedup(dg)
        n = min(len(cuts) - 1, 64)
        ids = np.arange(n, dtype=np.uint64)
        sig = O.minhash_chunks(data3, cuts, cfg, ids)
        keys, base = O.lsh(sig, cfg)
        dout, doff, kind = O.deflate_chunks(data3, cuts, cfg, ids, base)
        out[f"pipeline_3MiB_{name}"] = {
            "n_chunks": int(len(cuts) - 1), "cuts_first8": [int(c) for c in cuts[:8]], "cuts_sha256": h(cuts),
            "digests_sha256": h(dg), "first_occ_sha256": h(fo),
            "sig64_sha256": h(sig), "sig0_first4": [int(v) for v in sig[0, :4]], "band_keys64_sha256": h(keys),
            "base64": [int(b) for b in base], "deflate64_total": int(doff[-1]), "deflate64_sha256": h(dout),
            "kind64_delta": int((kind == 2).sum())}
    # near-duplicate family: one 200 KB article slice and 5 sparsely edited copies -> LSH bases and DELTA kinds
    var = variants_dataset(data3)
    out["variants_sha256"] = h(var)
    cfg = O.default_cfg(seg_size=1 << 20)
    cuts = O.cdc(var, cfg)
    dg = O.sha256_chunks(var, cuts)
    fo, rc = O.dedup(dg)
    uniq = np.nonzero(fo == np.arange(len(fo)))[0].astype(np.uint64)
    sig = O.minhash_chunks(var, cuts, cfg, uniq)
    keys, base = O.lsh(sig, cfg)
    dout, doff, kind = O.deflate_chunks(var, cuts, cfg, uniq, base)
    out["pipeline_variants"] = {"n_chunks": int(len(cuts) - 1), "n_unique": int(len(uniq)), "cuts_sha256": h(cuts),
                                "sig_sha256": h(sig), "base": [int(b) for b in base], "kind": [int(k) for k in kind],
                                "stream_len": [int(v) for v in np.diff(doff)], "streams_sha256": h(dout)}
    rnd = corpus.random_bytes(300_000)
    cfg = O.default_cfg()
    out["random_0xDEADBEEF_300k"] = {"sha256": h(rnd), "cuts_sha256": h(O.cdc(rnd, cfg)), "n_chunks": int(len(O.cdc(rnd, cfg)) - 1)}
    with open(os.path.join(HERE, "fixtures.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", os.path.join(HERE, "fixtures.json"))


if __name__ == "__main__":
    main()
