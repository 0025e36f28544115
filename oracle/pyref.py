"""Independent pure-Python restatement of L2 (Gear-FastCDC), L4a (MinHash) and L4b (LSH).

TEST INFRASTRUCTURE ONLY (same rules as oracle.py).  The reference ships no implementation or
golden vectors for these stages ("parity unpinned"), so the C oracle is pinned by
agreement with this second restatement, written from the same reference lines but sharing no code:
  L2  control flow README.md:2475-2490 (state never reset at a cut; size>=MIN && (hit || size>=MAX))
  L4a minhash_compute README.md:2578-2598 + public MurmurHash3_x86_32
  L4b banding README.md:1375-1383, 1987-1996
Pure-Python loops: small inputs only.
"""
from __future__ import annotations

import struct

M64 = (1 << 64) - 1
M32 = 0xFFFFFFFF


def gear_table():
    x = int.from_bytes(b"HMSE_L2G", "big")
    out = []
    for _ in range(256):
        x = (x + 0x9E3779B97F4A7C15) & M64
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        out.append(z ^ (z >> 31))
    return out


def masks(avg: int, norm: int):
    bits = avg.bit_length() - 1
    bs, bl = min(bits + norm, 48), max(bits - norm, 1)
    return (M64 << (64 - bs)) & M64, (M64 << (64 - bl)) & M64


def cdc(data: bytes, min_size=2048, avg_size=8192, max_size=32768, norm=2, seg=4 << 20):
    """Window form: the hash at byte i is sum_{k<64} G[b[i-k]] << k over bytes of the same segment
    (deliberately NOT the rolling recurrence the C oracle uses)."""
    G = gear_table()
    ms, ml = masks(avg_size, norm)
    cuts = [0]
    n = len(data)
    a = 0
    while a < n or (n == 0 and a == 0):
        b = min(a + seg, n)
        start = a
        i = a + min_size - 1
        while i < b:
            size = i + 1 - start
            h = 0
            for k in range(64):
                j = i - k
                if j < a:
                    break
                h = (h + (G[data[j]] << k)) & M64
            hit = (h & ms) == 0 if size < avg_size else (h & ml) == 0
            if hit or size >= max_size:
                cuts.append(i + 1)
                start = i + 1
                i = start + min_size - 1
            else:
                i += 1
        if start < b:
            cuts.append(b)
        a = b
        if n == 0:
            break
    return cuts


def murmur3_32(key: bytes, seed: int) -> int:
    c1, c2 = 0xCC9E2D51, 0x1B873593
    h = seed & M32
    nb = len(key) // 4
    for i in range(nb):
        (k,) = struct.unpack_from("<I", key, 4 * i)
        k = (k * c1) & M32
        k = ((k << 15) | (k >> 17)) & M32
        k = (k * c2) & M32
        h ^= k
        h = ((h << 13) | (h >> 19)) & M32
        h = (h * 5 + 0xE6546B64) & M32
    tail = key[4 * nb:]
    k = 0
    if len(tail) >= 3:
        k ^= tail[2] << 16
    if len(tail) >= 2:
        k ^= tail[1] << 8
    if len(tail) >= 1:
        k ^= tail[0]
        k = (k * c1) & M32
        k = ((k << 15) | (k >> 17)) & M32
        k = (k * c2) & M32
        h ^= k
    h ^= len(key)
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & M32
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & M32
    h ^= h >> 16
    return h


def minhash(data: bytes, n_hashes=128, shingle=4, seed_base=0):
    sig = [M32] * n_hashes
    if len(data) < shingle:
        return sig
    seen = set()
    for pos in range(len(data) - shingle + 1):
        seen.add(data[pos:pos + shingle])
    for sh in seen:  # min over a set == min over the multiset
        for h in range(n_hashes):
            v = murmur3_32(sh, seed_base + h)
            if v < sig[h]:
                sig[h] = v
    return sig


def lsh(sigs, bands=4, rows=32):
    """O(n^2) direct statement: base[i] = min j < i such that some whole band of sig j equals sig i's."""
    n = len(sigs)
    keys = [[murmur3_32(struct.pack("<%dI" % rows, *s[b * rows:(b + 1) * rows]), b) for b in range(bands)] for s in sigs]
    base = []
    for i in range(n):
        found = -1
        for j in range(i):
            if any(sigs[i][b * rows:(b + 1) * rows] == sigs[j][b * rows:(b + 1) * rows] for b in range(bands)):
                found = j
                break
        base.append(found)
    return keys, base
