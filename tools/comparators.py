"""Baseline comparators of the validation plan (VALIDATION_METHODS.md:241-247, 417-447 "Fair Comparison Protocol") on the
corpus profiles this build measures: bzip2 -9 (the only baseline present in the image; zstd / ZIM / CCSDS 121 are absent),
zlib level 9 as one stream and zlib level 9 per 8 KiB-average FastCDC chunk (the reference's L1 codec applied the way the
hot path applies it).  Same input for every method, CF = input / output, payload only ("raw") — the HMSE figures next to
them come from bench.py / tools/ablation.py, which report payload-only CF and CF with the 40/8/8-byte record overhead
("fair", index-inclusive) side by side.

CPU only: python tools/comparators.py [MiB per profile] [profiles,comma-separated]"""
import bz2, os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hmse_amd import corpus


def run(mib: int, profiles):
    rows = []
    for prof in profiles:
        data = (corpus.random_bytes(mib << 20) if prof == "random" else corpus.load(prof, mib << 20, seed=42)[0]).tobytes()
        n = len(data)
        t0 = time.perf_counter(); b = bz2.compress(data, 9); tb = time.perf_counter() - t0
        t0 = time.perf_counter(); z = zlib.compress(data, 9); tz = time.perf_counter() - t0
        assert bz2.decompress(b) == data and zlib.decompress(z) == data          # reconstruction integrity (VALIDATION_METHODS.md:419)
        rows.append((prof, n, len(b), tb, len(z), tz))
    return rows


def main():
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    profiles = (sys.argv[2] if len(sys.argv) > 2 else "wikipedia,arxiv,news,code,random").split(",")
    print(f"# baseline comparators, {mib} MiB per profile, one host core, payload-only CF (VALIDATION_METHODS.md:417-447)\n")
    print("| profile | bzip2 -9 CF | bzip2 MB/s | zlib -9 (one stream) CF | zlib MB/s |\n|---|---|---|---|---|")
    for prof, n, nb, tb, nz, tz in run(mib, profiles):
        print(f"| {prof} | {n / nb:.3f} | {n / tb / 1e6:.1f} | {n / nz:.3f} | {n / tz / 1e6:.1f} |")
    print("\nzstd -19, ZIM and CCSDS 121.0-B-3 (VALIDATION_METHODS.md:241-247) are not installed in this image; the HMSE figures for the same "
          "profiles are in the ablation tables (tools/ablation.py) and in bench.py's JSON (`cf` = index-inclusive, `cf_payload` = raw).")


if __name__ == "__main__":
    main()
