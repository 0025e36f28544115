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


# ---------------------------------------------------------------------------------------------
# L1: the build's DEFLATE encoder definition, restated independently of oracle/hmse_oracle_deflate.c
# (written from the six numbered rules in that file's header and from RFC 1951, sharing no code with
# it: dictionaries of bucket lists instead of a counting sort, Python deques for the two-queue
# Huffman, one big integer as bit buffer).  tests/test_oracle.py holds the two together byte for
# byte on small inputs; the definition itself stays "parity unpinned" at the reference (README.md
# names miniz level 9, whose serial lazy matcher a parallel encoder cannot reproduce: SURVEY.md §7).
# ---------------------------------------------------------------------------------------------
_LEVEL_DEPTH = (2, 2, 3, 4, 6, 8, 12, 16, 24, 32)
_CL_ORDER = (16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15)
# RFC 1951 §3.2.5 tables, written out (the C oracle uses closed forms)
_LEN_BASE = (3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258)
_LEN_EXTRA = (0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0)
_DIST_BASE = (1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
              8193, 12289, 16385, 24577)
_DIST_EXTRA = (0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13)


def _len_code(length):
    for c in range(28, -1, -1):
        if length >= _LEN_BASE[c]:
            return 257 + c, _LEN_EXTRA[c], length - _LEN_BASE[c]
    raise ValueError(length)


def _dist_code(dist):
    for c in range(29, -1, -1):
        if dist >= _DIST_BASE[c]:
            return c, _DIST_EXTRA[c], dist - _DIST_BASE[c]
    raise ValueError(dist)


def lz_anchors(w: bytes, dl: int, buckets: dict) -> dict:
    """Rule 2b, first half: block a of a dictionary job's chunk -> (distance, run) of its anchor position dl + 64 a."""
    t = len(w)
    padded = w + bytes(64)
    out = {}
    for a in range((t - dl + 63) // 64):
        pa = dl + 64 * a
        if pa + 4 > t:
            continue
        h = ((int.from_bytes(w[pa:pa + 4], "little") * 0x9E3779B1) & M32) >> 20
        near = [q for q in buckets.get(h, []) if q < dl][-64:]          # the nearest 64 dictionary members
        score = {}
        for q in near:
            if pa - q > 32768:
                continue
            n = 0
            while n < 32 and padded[q + n] == padded[pa + n]:
                n += 1
            score[q] = n
        if not score:
            continue
        top = max(score.values())
        if top < 32:
            continue
        q = max(q for q, n in score.items() if n == top)                  # ties: nearest
        d, run = pa - q, 0
        while run < min(512, t - pa) and w[pa + run] == w[pa + run - d]:
            run += 1
        out[a] = (d, run)
    return out


def lz_matches(chunk: bytes, dict_: bytes = b"", depth: int = 32):
    """Rules 1-2c: per chunk position the longest match among its `depth` nearest bucket predecessors; in a dictionary job a
    position inside an anchor's diagonal run (>= 16 bytes of it left) takes that match outright (no walk)."""
    dict_ = dict_[-32768:]
    w = dict_ + chunk
    t, dl = len(w), len(dict_)
    buckets = {}
    for q in range(t - 3):
        h = ((int.from_bytes(w[q:q + 4], "little") * 0x9E3779B1) & M32) >> 20
        buckets.setdefault(h, []).append(q)
    where = {}
    for members in buckets.values():
        for r, q in enumerate(members):
            where[q] = (members, r)
    anchors = lz_anchors(w, dl, buckets) if dl else {}
    mlen, mdist = [0] * len(chunk), [0] * len(chunk)
    for p in range(dl, t - 3):
        members, r = where[p]
        cap = min(258, t - p)
        hint = (0, 0)                                                     # (known length, candidate position)
        for a in ((p - dl) // 64, (p - dl) // 64 - 1):
            if a < 0 or a not in anchors:
                continue
            d, run = anchors[a]
            pa = dl + 64 * a
            if d > p or p - d >= dl:
                continue
            m = cap if run >= 512 else (min(pa + run - p, cap) if pa + run > p else 0)
            if m > hint[0]:
                hint = (m, p - d)
        best, bdist = 3, 0
        if hint[0] >= 16:                                      # rule 2c: a hinted match is taken outright
            best, bdist = hint[0], p - hint[1]
        else:
            for q in reversed(members[max(0, r - depth):r]):   # nearest first
                if p - q > 32768:
                    break
                n = 0
                while n < cap and w[q + n] == w[p + n]:
                    n += 1
                if n > best:
                    best, bdist = n, p - q
                    if n == cap:
                        break
        if best >= 4:
            mlen[p - dl], mdist[p - dl] = best, bdist
    return mlen, mdist


def lz_parse(chunk: bytes, mlen, mdist):
    """Rule 3: one-step lazy parse -> list of (literal byte, 0) / (length, distance)."""
    toks, p, n = [], 0, len(chunk)
    while p < n:
        here = mlen[p]
        nxt = mlen[p + 1] if p + 1 < n else 0
        if here >= 4 and not nxt > here:
            toks.append((here, mdist[p]))
            p += here
        else:
            toks.append((chunk[p], 0))
            p += 1
    return toks


def huffman_lengths(freq, limit):
    """Rule 5: two-queue Huffman over (freq, index)-sorted leaves, leaf wins ties; depths clamped to `limit`, Kraft repair,
    lengths handed out in sorted order (most frequent = shortest)."""
    from collections import deque
    freq = list(freq)
    for i in range(len(freq)):             # fewer than two used symbols: the lowest unused indices count once
        if sum(1 for f in freq if f) >= 2:
            break
        if not freq[i]:
            freq[i] = 1
    leaves = sorted((f, s) for s, f in enumerate(freq) if f)
    m = len(leaves)
    # nodes: (weight, [leaf ranks below it]); depth of a leaf = number of merges it took part in
    depth = [0] * m
    lq = deque((f, [r]) for r, (f, _) in enumerate(leaves))
    iq = deque()
    while len(lq) + len(iq) > 1:
        picked = []
        for _ in range(2):
            if lq and (not iq or lq[0][0] <= iq[0][0]):
                picked.append(lq.popleft())
            else:
                picked.append(iq.popleft())
        below = picked[0][1] + picked[1][1]
        for r in below:
            depth[r] += 1
        iq.append((picked[0][0] + picked[1][0], below))
    count = [0] * (limit + 1)
    for d in depth:
        count[min(d, limit)] += 1
    total = sum(count[i] << (limit - i) for i in range(1, limit + 1))
    while total > (1 << limit):            # over-subscribed after clamping: miniz-style repair
        count[limit] -= 1
        for i in range(limit - 1, 0, -1):
            if count[i]:
                count[i] -= 1
                count[i + 1] += 2
                break
        total -= 1
    lens = [0] * len(freq)
    rank = m
    for length in range(1, limit + 1):
        for _ in range(count[length]):
            rank -= 1
            lens[leaves[rank][1]] = length
    return lens


def canonical_codes(lens):
    """RFC 1951 §3.2.2; returned bit-reversed (DEFLATE packs Huffman codes MSB first into an LSB-first stream)."""
    per_len = [0] * 16
    for n in lens:
        if n:
            per_len[n] += 1
    nxt, code = [0] * 16, 0
    for bits in range(1, 16):
        code = (code + per_len[bits - 1]) << 1
        nxt[bits] = code
    out = []
    for n in lens:
        if not n:
            out.append(0)
            continue
        c = nxt[n]
        nxt[n] += 1
        out.append(int(format(c, "0%db" % n)[::-1], 2))
    return out


def rle_code_lengths(lens):
    """Rule 6 for one tree -> list of (symbol, extra bits, extra value)."""
    out, i, n = [], 0, len(lens)
    while i < n:
        j = i
        while j < n and lens[j] == lens[i]:
            j += 1
        run, v = j - i, lens[i]
        if v == 0:
            while run >= 11:
                c = min(run, 138)
                out.append((18, 7, c - 11))
                run -= c
            if run >= 3:
                out.append((17, 3, run - 3))
                run = 0
            out.extend([(0, 0, 0)] * run)
        else:
            out.append((v, 0, 0))
            run -= 1
            while run >= 3:
                c = min(run, 6)
                out.append((16, 2, c - 3))
                run -= c
            out.extend([(v, 0, 0)] * run)
        i = j
    return out


class _Bits:
    def __init__(self):
        self.acc, self.n = 0, 0

    def put(self, value, nbits):
        self.acc |= value << self.n
        self.n += nbits

    def align(self):
        self.n = (self.n + 7) & ~7

    def bytes(self):
        self.align()
        return self.acc.to_bytes(self.n // 8, "little")


def deflate(chunk: bytes, dict_: bytes = b"", level: int = 9, chain_depth: int = 0) -> bytes:
    """The whole encoder (rules 1-6): one final block, smallest of stored / fixed / dynamic (ties: stored, then fixed)."""
    depth = chain_depth or _LEVEL_DEPTH[min(level, 9)]
    mlen, mdist = lz_matches(chunk, dict_, depth)
    toks = lz_parse(chunk, mlen, mdist)
    lf, df, extra = [0] * 286, [0] * 30, 0
    for a, d in toks:
        if d:
            c, eb, _ = _len_code(a)
            lf[c] += 1
            extra += eb
            c, eb, _ = _dist_code(d)
            df[c] += 1
            extra += eb
        else:
            lf[a] += 1
    lf[256] += 1
    fixed_ll = [8] * 144 + [9] * 112 + [7] * 24 + [8] * 8
    fixed_bits = 3 + extra + sum(f * fixed_ll[s] for s, f in enumerate(lf)) + 5 * sum(df)
    ll, dl = huffman_lengths(lf, 15), huffman_lengths(df, 15)
    nlit = max(257, max(s + 1 for s in range(286) if ll[s]))
    ndist = max(1, max((s + 1 for s in range(30) if dl[s]), default=1))
    rle = rle_code_lengths(ll[:nlit]) + rle_code_lengths(dl[:ndist])
    cf = [0] * 19
    for s, _, _ in rle:
        cf[s] += 1
    cl = huffman_lengths(cf, 7)
    ncl = max(4, max(i + 1 for i in range(19) if cl[_CL_ORDER[i]]))
    dyn_bits = 3 + 14 + 3 * ncl + extra + sum(cl[s] + eb for s, eb, _ in rle) \
        + sum(f * ll[s] for s, f in enumerate(lf)) + sum(f * dl[s] for s, f in enumerate(df))
    stored_bits = 8 * (5 + len(chunk))
    b = _Bits()
    if stored_bits <= fixed_bits and stored_bits <= dyn_bits:
        b.put(1, 1); b.put(0, 2); b.align()
        b.put(len(chunk) & 0xFFFF, 16); b.put(~len(chunk) & 0xFFFF, 16)
        return b.bytes() + chunk
    if fixed_bits <= dyn_bits:
        ll, dl = fixed_ll, [5] * 32
        lc, dc = canonical_codes(ll), canonical_codes(dl)
        b.put(1, 1); b.put(1, 2)
    else:
        lc, dc, cc = canonical_codes(ll), canonical_codes(dl), canonical_codes(cl)
        b.put(1, 1); b.put(2, 2)
        b.put(nlit - 257, 5); b.put(ndist - 1, 5); b.put(ncl - 4, 4)
        for i in range(ncl):
            b.put(cl[_CL_ORDER[i]], 3)
        for s, eb, ev in rle:
            b.put(cc[s], cl[s])
            b.put(ev, eb)
    for a, d in toks:
        if d:
            c, eb, ev = _len_code(a)
            b.put(lc[c], ll[c]); b.put(ev, eb)
            c, eb, ev = _dist_code(d)
            b.put(dc[c], dl[c]); b.put(ev, eb)
        else:
            b.put(lc[a], ll[a])
    b.put(lc[256], ll[256])
    return b.bytes()


def deflate_record(chunk: bytes, base: bytes | None = None, level: int = 9, chain_depth: int = 0, delta_max_ratio_pct: int = 0):
    """Rule 7 — FULL or DELTA (README.md:1328, 2175): with a base the dictionary stream comes first and IS the record
    when it is at most a fifth of the chunk (the FULL stream is then never computed); a larger one is kept iff it nets savings over
    FULL after the 8-byte DeltaChunk header.  Returns (stream, kind) with kind 0 = FULL, 2 = DELTA."""
    if base is None:
        return deflate(chunk, b"", level, chain_depth), 0
    d = deflate(chunk, base, level, chain_depth)
    gate = not (delta_max_ratio_pct and len(d) * 100 > delta_max_ratio_pct * len(chunk))
    if gate and 5 * len(d) <= len(chunk):
        return d, 2
    f = deflate(chunk, b"", level, chain_depth)
    return (d, 2) if gate and len(d) + 8 < len(f) else (f, 0)
