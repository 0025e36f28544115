"""DEFLATE chain-depth sweep: match-kernel time against stored size, with zlib level 9 on the same chunks
as the reference point (the codec the reference names, README.md:2374).

    python tools/depth_sweep.py [MiB per profile] [depths, comma-separated]

Per corpus profile (+ incompressible bytes): one ingest for cuts / stored chunks / LSH bases (none depends on the depth), zlib-9
sizes of every stored chunk (FULL, and with zdict = base where a base exists) on host threads, then hmse_l1_deflate per depth:
  plain  = every stored chunk without a dictionary (FULL sizes, plain-class match kernels)
  dict   = the product call (bases given: DELTA where it nets savings)
Columns: match ms plain / match ms product (plain + dictionary classes) / encode ms / FULL bytes vs zlib-9 / stored bytes vs zlib-9.
"""
import ctypes as C
import os
import sys
import zlib
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hmse_amd import IngestConfig, _lib, corpus, ingest, ops

MIB = int(sys.argv[1]) if len(sys.argv) > 1 else 256
DEPTHS = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "4,8,12,16,24,32").split(",")]
dev = torch.device("cuda:0")
lib = _lib.hip_lib()
MATCH_PLAIN, MATCH_DICT, ENC = range(8, 14), range(18, 24), (14, 15)


def slots_ms(slots):
    tot = 0.0
    for s in slots:
        ms, n = C.c_double(), C.c_uint64()
        lib.hmse_profile_read(s, C.byref(ms), C.byref(n), 1)
        tot += ms.value
    return tot


def zlib_sizes(host, cuts, uniq, base):
    def one(k):
        c = int(uniq[k])
        ch = host[int(cuts[c]):int(cuts[c + 1])].tobytes()
        co = zlib.compressobj(9, zlib.DEFLATED, -15, 9, zlib.Z_DEFAULT_STRATEGY)
        full = len(co.compress(ch) + co.flush())
        delta = -1
        if base[k] >= 0:
            b = int(uniq[base[k]])
            co = zlib.compressobj(9, zlib.DEFLATED, -15, 9, zlib.Z_DEFAULT_STRATEGY, host[int(cuts[b]):int(cuts[b + 1])].tobytes())
            delta = len(co.compress(ch) + co.flush())
        return full, delta
    with ThreadPoolExecutor(16) as ex:
        r = list(ex.map(one, range(len(uniq)), chunksize=256))
    full = np.array([x[0] for x in r], np.int64)
    delta = np.array([x[1] for x in r], np.int64)
    raw = np.array([int(cuts[int(c) + 1] - cuts[int(c)]) for c in uniq], np.int64)
    stored = np.where((delta >= 0) & ((5 * delta <= raw) | (delta + 8 < full)), delta, full)
    return int(full.sum()), int(stored.sum())


def main():
    print(f"# depth sweep, {MIB} MiB per profile, depths {DEPTHS}; zlib {zlib.ZLIB_RUNTIME_VERSION}")
    print("profile,depth,match_plain_only_ms,match_product_ms(plain+dict),encode_product_ms,full_bytes,full_vs_zlib9_pct,stored_bytes,stored_vs_zlib9_pct,delta_records")
    for prof in ("wikipedia", "arxiv", "news", "code", "random"):
        n = (MIB if prof != "random" else min(MIB, 64)) << 20
        if prof == "random":
            host = np.random.default_rng(0xDEADBEEF).integers(0, 256, n, dtype=np.uint8)
        else:
            host = corpus.load(prof, n, seed=42)[0]
        data = torch.from_numpy(host).to(dev)
        cfg0 = IngestConfig()
        res = ingest.ingest_shard(data, cfg0)
        cuts, uniq, base = res.cuts.cpu().numpy(), res.uniq_ids.cpu().numpy(), res.base.cpu().numpy()
        zfull, zstored = zlib_sizes(host, cuts, uniq, base)
        for d in DEPTHS:
            cfg = cfg0.with_(chain_depth=d)
            for warm in (True, False):
                lib.hmse_profile_enable(0 if warm else 1)
                for s in list(MATCH_PLAIN) + list(MATCH_DICT) + list(ENC):
                    lib.hmse_profile_read(s, None, None, 1)
                out_f, _, _ = ops.l1_deflate(data, res.cuts, cfg, res.uniq_ids, None)
                torch.cuda.synchronize()
                ms_plain = slots_ms(MATCH_PLAIN)
                slots_ms(MATCH_DICT); slots_ms(ENC)
                out_p, _, kind = ops.l1_deflate(data, res.cuts, cfg, res.uniq_ids, res.base)
                torch.cuda.synchronize()
                ms_prod = slots_ms(MATCH_PLAIN) + slots_ms(MATCH_DICT)
                ms_enc = slots_ms(ENC)
            lib.hmse_profile_enable(0)
            fb, sb = int(out_f.numel()), int(out_p.numel())
            print(f"{prof},{d},{ms_plain:.2f},{ms_prod:.2f},{ms_enc:.2f},{fb},{100.0 * (fb - zfull) / zfull:+.3f},{sb},{100.0 * (sb - zstored) / zstored:+.3f},"
                  f"{int((kind == 2).sum())}", flush=True)
            del out_f, out_p
        del data, res


if __name__ == "__main__":
    main()
