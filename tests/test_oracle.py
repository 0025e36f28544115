"""CPU: the oracle against standards, the reference's own arithmetic, an independent restatement and the
committed fixtures.  No GPU.  (Parity pinning status: oracle/hmse_oracle.h.)"""
import hashlib
import json
import math
import os
import zlib

import numpy as np
import pytest

from conftest import words_text

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = json.load(open(os.path.join(HERE, "golden", "fixtures.json")))


def h(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ---------------------------------------------------------------- L3: FIPS 180-4
def test_sha256_standard_vectors(orc):
    assert orc.sha256(b"abc").hex() == "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad"
    assert orc.sha256(b"").hex() == "e3b0c44298fc1c149afbf4c8996fb92427ae41e4649b934ca495991b7852b855"
    assert orc.sha256(b"abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq").hex() == \
        "248d6a61d20638b8e5c026930c3e6039a33ce45964ff2167f6ecedd419db06c1"


def test_sha256_vs_hashlib_all_padding_branches(orc):
    rng = np.random.default_rng(0)
    for n in list(range(0, 130)) + [4095, 4096, 8191, 65536]:
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert orc.sha256(d) == hashlib.sha256(d).digest(), n


# ---------------------------------------------------------------- L4a: MurmurHash3_x86_32 known answers
MURMUR_KATS = [
    (b"", 0, 0x00000000), (b"", 1, 0x514E28B7), (b"", 0xFFFFFFFF, 0x81F16F39), (b"\xff\xff\xff\xff", 0, 0x76293B50),
    (b"\x21\x43\x65\x87", 0, 0xF55B516B), (b"\x21\x43\x65\x87", 0x5082EDEE, 0x2362F9DE), (b"\x21\x43\x65", 0, 0x7E4A8634),
    (b"\x21\x43", 0, 0xA0F7B07A), (b"\x21", 0, 0x72661CF4), (b"\0\0\0\0", 0, 0x2362F9DE), (b"\0\0\0", 0, 0x85F0B427),
    (b"\0\0", 0, 0x30F4C306), (b"\0", 0, 0x514E28B7), (b"test", 0, 0xBA6BD213), (b"Hello, world!", 0, 0xC0363E43),
    (b"The quick brown fox jumps over the lazy dog", 0, 0x2E4FF723)]


def test_murmur3_known_answers(orc):
    from oracle import pyref
    for key, seed, want in MURMUR_KATS:
        assert orc.murmur3(key, seed) == want
        assert pyref.murmur3_32(key, seed) == want


# ---------------------------------------------------------------- L2
def test_gear_table_fixture_and_restatement(orc):
    from oracle import pyref
    g = orc.gear_table()
    assert list(g) == pyref.gear_table()
    assert h(g) == FIX["gear_table"]["sha256"]
    assert [int(x) for x in g[:4]] == FIX["gear_table"]["first4"] and int(g[255]) == FIX["gear_table"]["last"]
    assert len(set(g.tolist())) == 256


def test_masks(orc):
    from oracle import pyref
    ms, ml = orc.cdc_masks(orc.default_cfg())
    assert [f"{ms:#018x}", f"{ml:#018x}"] == FIX["masks_default"] and (ms, ml) == pyref.masks(8192, 2)
    assert ml & ms == ml  # the easy mask is a subset of the hard one: S-candidates are L-candidates
    assert bin(ms).count("1") == 15 and bin(ml).count("1") == 11


@pytest.mark.parametrize("kind", ["random", "words", "zeros", "short", "empty"])
def test_cdc_matches_independent_restatement(orc, kind):
    """C oracle (rolling recurrence) == pure-Python window form (sum_{k<64} G[b[i-k]] << k)."""
    from oracle import pyref
    rng = np.random.default_rng(42)
    data = {"random": rng.integers(0, 256, 40000, dtype=np.uint8), "words": words_text(40000),
            "zeros": np.zeros(9000, np.uint8), "short": words_text(100), "empty": np.zeros(0, np.uint8)}[kind]
    cfg = orc.default_cfg(min_size=64, avg_size=256, max_size=1024, seg_size=5000)
    assert list(orc.cdc(data, cfg)) == pyref.cdc(data.tobytes(), 64, 256, 1024, 2, 5000)


def test_cdc_bounds_segments_and_determinism(orc, corpus_small):
    cfg = orc.default_cfg()
    cuts = orc.cdc(corpus_small, cfg)
    assert np.array_equal(cuts, orc.cdc(corpus_small, cfg))          # reruns bitwise identical (VALIDATION_METHODS.md:409)
    sz = np.diff(cuts.astype(np.int64))
    assert cuts[0] == 0 and cuts[-1] == corpus_small.size and (sz > 0).all()
    assert sz.max() <= cfg.max_size
    seg_ends = set(range(cfg.seg_size, corpus_small.size, cfg.seg_size)) | {corpus_small.size}
    short = [int(cuts[i + 1]) for i in range(len(sz)) if sz[i] < cfg.min_size]
    assert all(e in seg_ends for e in short)                          # only a segment's tail chunk may be < MIN
    assert all(e in set(cuts.tolist()) for e in seg_ends)             # forced cut at every segment end
    assert 6000 < sz.mean() < 12000                                   # 8 KiB target (README.md:2512 band, scaled)


def test_cdc_shift_resistance(orc, corpus_small):
    """Insert 100 bytes at the start -> >= 99 % of chunks keep their digest (README.md:1254)."""
    cfg = orc.default_cfg(seg_size=1 << 30)
    a = corpus_small[: 4_000_000]
    b = np.concatenate([np.frombuffer(os.urandom(100), np.uint8), a])
    da = {orc.sha256(a[int(s):int(e)]) for s, e in zip(orc.cdc(a, cfg)[:-1], orc.cdc(a, cfg)[1:])}
    cb = orc.cdc(b, cfg)
    db = [orc.sha256(b[int(s):int(e)]) for s, e in zip(cb[:-1], cb[1:])]
    assert sum(d in da for d in db) / len(db) >= 0.99


def test_reference_skeleton_acceptance_band(orc):
    """The literal skeleton (README.md:2456-2490) on ASCII words: min >= 1024, max <= 16384 (README.md:2513-2514);
    its average is MIN + 4096-ish, not the 3.5-4.5 KB the README hopes for."""
    d = words_text(3_000_000)
    sz = np.diff(orc.cdc_reference_skeleton(d).astype(np.int64))
    assert sz[:-1].min() >= 1024 and sz.max() <= 16384
    assert 3500 < sz.mean() < 6000


# ---------------------------------------------------------------- L4
def test_minhash_matches_restatement_and_guards(orc):
    from oracle import pyref
    cfg = orc.default_cfg()
    t = words_text(3000).tobytes()
    assert list(orc.minhash(t, cfg)) == pyref.minhash(t)
    c1 = orc.default_cfg(seed_base=1)                                  # VALIDATION_METHODS.md:122 seeds 1..128
    assert list(orc.minhash(t, c1)) == pyref.minhash(t, seed_base=1)
    assert list(orc.minhash(t, c1))[:127] == list(orc.minhash(t, cfg))[1:]
    for short in (b"", b"a", b"abc"):                                  # len < 4: size_t underflow guarded (README.md:2585)
        assert (orc.minhash(short, cfg) == 0xFFFFFFFF).all()
    assert list(orc.minhash(b"abcd", cfg)) == [pyref.murmur3_32(b"abcd", s) for s in range(128)]


def test_minhash_estimates_jaccard(orc):
    """Signature agreement ~ Jaccard of the shingle sets (README.md:1359-1373)."""
    cfg = orc.default_cfg()
    rng = np.random.default_rng(7)
    a = words_text(8000, seed=1)
    b = a.copy(); b[rng.integers(0, 8000, 60)] = 35
    sa, sb = orc.minhash(a, cfg), orc.minhash(b, cfg)
    sh = lambda x: {x[i:i + 4].tobytes() for i in range(len(x) - 3)}
    A, B = sh(a), sh(b)
    j = len(A & B) / len(A | B)
    assert abs((sa == sb).mean() - j) < 0.12


def test_lsh_matches_bruteforce_and_formula(orc):
    from oracle import pyref
    cfg = orc.default_cfg()
    rng = np.random.default_rng(9)
    n = 400
    sig = rng.integers(0, 2**32, (n, 128), dtype=np.uint64).astype(np.uint32)
    for i in range(50, n):
        if rng.random() < 0.4:
            j = int(rng.integers(0, i)); b = int(rng.integers(0, 4))
            sig[i, 32 * b:32 * b + 32] = sig[j, 32 * b:32 * b + 32]
    keys, base = orc.lsh(sig, cfg)
    pk, pb = pyref.lsh([list(map(int, s)) for s in sig])
    assert base.tolist() == pb and keys.tolist() == pk
    # S-curve: P(candidate) = 1-(1-s^r)^b (README.md:2233-2235) — assert the FORMULA, never the README table
    b_, r_ = 4, 32
    for s, want in ((0.85, 0.0219), (0.90, 0.130), (0.95, 0.577)):
        assert abs(1 - (1 - s ** r_) ** b_ - want) < 2e-3
    trials, hit = 3000, 0
    s = 0.95
    x = rng.integers(0, 2**32, (trials, 128), dtype=np.uint64).astype(np.uint32)
    y = x.copy()
    flip = rng.random((trials, 128)) > s
    y[flip] ^= 1
    both = np.concatenate([x, y])
    _, bb = orc.lsh(both, cfg)
    hit = (bb[trials:] == np.arange(trials)).mean()
    assert abs(hit - (1 - (1 - s ** r_) ** b_)) < 0.04


# ---------------------------------------------------------------- L1
def _rt(stream, zdict=None):
    d = zlib.decompressobj(-15, zdict=zdict) if zdict else zlib.decompressobj(-15)
    out = d.decompress(stream) + d.flush()
    assert d.eof and not d.unused_data
    return out


def test_deflate_roundtrip_edges(orc):
    cfg = orc.default_cfg()
    rng = np.random.Generator(np.random.PCG64(0xDEADBEEF))
    cases = [b"", b"a", b"ab", b"abc", b"abcd", b"aaaa", b"abcabcabcabcabc", bytes(32768), bytes([7]) * 1000,
             rng.integers(0, 256, 10000, dtype=np.uint8).tobytes(), words_text(32768).tobytes(),
             np.tile(np.arange(256, dtype=np.uint8), 100).tobytes(), rng.integers(0, 2, 20000, dtype=np.uint8).tobytes()]
    for lvl in (1, 5, 9):
        c = orc.default_cfg(level=lvl)
        for x in cases:
            s = orc.deflate(x, c)
            assert _rt(s) == x
            assert len(s) <= len(x) + 5                                # never worse than one stored block
    r = rng.integers(0, 256, 5000, dtype=np.uint8).tobytes()
    assert len(orc.deflate(r, cfg)) == 5005                            # incompressible -> stored, CF ~ 1 (VALIDATION_METHODS.md:213)
    assert orc.deflate(b"", cfg) == b"\x03\x00"


def test_deflate_dictionary_and_delta_rule(orc):
    cfg = orc.default_cfg()
    base = words_text(9000, seed=3).tobytes()
    var = bytearray(base); var[100:108] = b"XXXXXXXX"; var = bytes(var)
    full, delta = orc.deflate(var, cfg), orc.deflate(var, cfg, base)
    assert _rt(delta, base) == var and len(delta) * 10 < len(full)
    data = np.frombuffer(base + var + os.urandom(3000), np.uint8)
    cuts = np.array([0, 9000, 18000, 21000], np.uint64)
    out, off, kind = orc.deflate_chunks(data, cuts, cfg, None, np.array([-1, 0, 0], np.int64))
    assert kind.tolist() == [0, 2, 0]                                  # DELTA only when it nets savings
    assert _rt(out[int(off[1]):int(off[2])].tobytes(), base) == var
    c20 = orc.default_cfg(delta_max_ratio_pct=20)
    assert orc.deflate_chunks(data, cuts, c20, None, np.array([-1, 0, 0], np.int64))[2].tolist() == [0, 2, 0]


def test_deflate_cf_close_to_zlib9(orc, corpus_small):
    """CF sanity: >= 2:1 on Wikipedia-like text (README.md:2425) and what DESIGN.md §2 states about zlib level 9 (the
    codec README.md:2374 names): FULL streams are never more than 0.3 % larger in total than zlib-9's on the same chunks
    (measured: 1.9-2.5 % smaller on the corpus profiles and on plain word text)."""
    cfg = orc.default_cfg()
    for data in (corpus_small[: 1 << 20], words_text(1 << 19, seed=3)):
        cuts = orc.cdc(data, cfg)
        mine = z9 = 0
        for s, e in zip(cuts[:-1], cuts[1:]):
            c = data[int(s):int(e)].tobytes()
            mine += len(orc.deflate(c, cfg))
            co = zlib.compressobj(9, zlib.DEFLATED, -15, 9)
            z9 += len(co.compress(c) + co.flush())
        assert data.size / mine >= 2.0
        assert mine <= 1.003 * z9


def test_delta_records_close_to_zlib9_with_zdict(orc):
    """DELTA records (raw DEFLATE with zdict = base chunk) against zlib level 9 given the same dictionary, on
    the near-duplicate family of the golden fixtures: every record inflates through stock zlib with that dictionary, is
    many times smaller than the FULL record, and the DELTA bytes in total stay within 5 % of zlib's (measured +2.3 % on
    24 MiB of wiki-synth, +4.4 % here: ~10 bytes on records of ~250 bytes; depth 32 against zlib's 4096-deep chains)."""
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from make_golden import variants_dataset
    from hmse_amd import corpus
    cfg = orc.default_cfg(seg_size=1 << 20)
    data = variants_dataset(corpus.wiki_synth(3 << 20, seed=42))
    cuts = orc.cdc(data, cfg)
    fo, _ = orc.dedup(orc.sha256_chunks(data, cuts))
    uniq = np.nonzero(fo == np.arange(len(fo)))[0].astype(np.uint64)
    _, base = orc.lsh(orc.minhash_chunks(data, cuts, cfg, uniq), cfg)
    out, off, kind = orc.deflate_chunks(data, cuts, cfg, uniq, base)
    mine = zl = full = n = 0
    for k in np.nonzero(kind == 2)[0]:
        c = data[int(cuts[uniq[k]]):int(cuts[uniq[k] + 1])].tobytes()
        b = uniq[base[k]]
        d = data[int(cuts[b]):int(cuts[b + 1])].tobytes()
        rec = out[int(off[k]):int(off[k + 1])].tobytes()
        assert _rt(rec, d) == c
        co = zlib.compressobj(9, zlib.DEFLATED, -15, 9, zlib.Z_DEFAULT_STRATEGY, d)
        zl += len(co.compress(c) + co.flush())
        mine += len(rec)
        full += len(orc.deflate(c, cfg))
        n += 1
    assert n >= 20
    assert mine <= 1.05 * zl and mine * 8 < full


def test_deflate_c_oracle_equals_python_restatement(orc):
    """The encoder DEFINITION is this build's own (parity unpinned at the reference): it is pinned here by two
    restatements that share no code — oracle/hmse_oracle_deflate.c and oracle/pyref.py — agreeing byte for byte on match
    selection, lazy parse, block choice, Huffman lengths, code-length RLE and bit packing, over inputs that reach every
    branch: tiny inputs, one-symbol runs, 2-/4-symbol alphabets (length-limited trees), incompressible bytes (stored),
    barely compressible bytes (stored / fixed / dynamic ties), text, dictionaries, every level's depth."""
    from oracle import pyref
    rng = np.random.Generator(np.random.PCG64(99))
    text = words_text(6000, seed=5).tobytes()
    cases = [b"", b"a", b"ab", b"abc", b"abcd", b"abcde", b"aaaa", b"abcabcabcabc", bytes(300), bytes(3000), b"x" * 259, b"xy" * 700,
             bytes(range(256)) * 3, text[:1], text[:37], text[:600], text[:2500], rng.integers(0, 256, 1200, dtype=np.uint8).tobytes(),
             rng.integers(0, 2, 3000, dtype=np.uint8).tobytes(), rng.integers(0, 4, 2500, dtype=np.uint8).tobytes(),
             rng.integers(0, 16, 1800, dtype=np.uint8).tobytes(), rng.integers(0, 250, 900, dtype=np.uint8).tobytes()]
    for L in (700, 1300):   # random bytes with a repeated tail: sizes around the stored / dynamic break-even
        body = rng.integers(0, 256, L, dtype=np.uint8)
        for t in (20, 40, 48, 52, 56, 64, 90):
            c = body.copy(); c[L - t:] = c[:t]
            cases.append(c.tobytes())
    # skewed frequencies: Fibonacci-like counts force depths beyond the 15-bit (and 7-bit code-length) limits -> Kraft repair
    fib, skew = [1, 1], bytearray()
    while len(fib) < 22:
        fib.append(fib[-1] + fib[-2])
    for sym, f in enumerate(fib):
        skew += bytes([65 + sym]) * min(f, 4000)
    perm = rng.permutation(len(skew))
    cases.append(bytes(np.frombuffer(bytes(skew), np.uint8)[perm][:9000]))
    block = {0: 0, 1: 0, 2: 0}
    for c in cases:
        for lvl in (9, 1):
            cfg = orc.default_cfg(level=lvl)
            want, got = orc.deflate(c, cfg), pyref.deflate(c, level=lvl)
            assert want == got, (len(c), lvl)
            block[(got[0] >> 1) & 3] += 1
            assert _rt(got) == c
    assert all(block.values()), block   # stored, fixed and dynamic blocks all occurred
    var = bytearray(text[:2500]); var[100:104] = b"ZZZZ"; var[900:901] = b""; var = bytes(var)
    for c, d in ((var, text[:2500]), (text[600:1800], text[:2500]), (text[:300], bytes(200)), (b"abc", b"abcabc"),
                 (rng.integers(0, 256, 500, dtype=np.uint8).tobytes(), text[:700])):
        for kw in (dict(), dict(chain_depth=3), dict(chain_depth=200)):
            cfg = orc.default_cfg(**kw)
            want, got = orc.deflate(c, cfg, d), pyref.deflate(c, d, chain_depth=kw.get("chain_depth", 0))
            assert want == got, (len(c), len(d), kw)
            assert _rt(got, d) == c
    # dictionary jobs with insertions / deletions / substitutions: several diagonals, anchors and hints of rule 2b
    base = words_text(7000, seed=9).tobytes()
    erng = np.random.default_rng(3)
    for trial in range(6):
        v = bytearray(base)
        for _ in range(int(erng.integers(1, 6))):
            pos = int(erng.integers(0, len(v) - 50)); kind = int(erng.integers(0, 3))
            if kind == 0:
                v[pos:pos] = bytes(erng.integers(97, 123, int(erng.integers(1, 40)), dtype=np.uint8))
            elif kind == 1:
                del v[pos:pos + int(erng.integers(1, 40))]
            else:
                v[pos] = 63
        v = bytes(v)
        want, got = orc.deflate(v, orc.default_cfg(), base), pyref.deflate(v, base)
        assert want == got, trial
        assert _rt(got, base) == v and len(got) * 6 < len(v)
    ml, md = orc.deflate_matches(text[:2500].encode() if isinstance(text, str) else text[:2500], orc.default_cfg())
    pl, pd = pyref.lz_matches(text[:2500], b"", 32)
    assert ml.tolist() == pl and md.tolist() == pd
    # rule 2c (round 3): a position with a full-length diagonal hint takes it outright — also where the chunk itself holds a
    # NEARER full-length candidate (a repeat inside the chunk), which rule 2b preferred
    x = erng.integers(0, 256, 900, dtype=np.uint8).tobytes()
    rep = x + x + base[:1500]
    drep = bytearray(rep); drep[40] ^= 1; drep = bytes(drep)
    want, got = orc.deflate(rep, orc.default_cfg(), drep), pyref.deflate(rep, drep)
    assert want == got and _rt(got, drep) == rep
    ml, md = orc.deflate_matches(rep, orc.default_cfg(), drep)
    pl, pd = pyref.lz_matches(rep, drep, 32)
    assert ml.tolist() == pl and md.tolist() == pd
    p2 = 900 + 300                                      # inside the second copy of x: chunk candidate at distance 900, dictionary at len(drep)
    assert int(ml[p2]) == 258 and int(md[p2]) == len(drep)


def test_record_kind_rule_c_oracle_equals_python_restatement(orc):
    """Rule 7 (README.md:1328, 2175): a delta of at most a fifth of the chunk is the record and FULL is never
    computed; a larger one must net savings over FULL after the 8-byte header; the optional percentage gate refuses first.
    orc_deflate_chunks against pyref.deflate_record on near-duplicates, an unrelated base, a base that does not help a highly
    compressible chunk, and the 20 % gate."""
    from oracle import pyref
    rng = np.random.default_rng(17)
    t1 = words_text(5000, seed=21).tobytes()
    v1 = bytearray(t1); v1[700:705] = b"#####"; v1 = bytes(v1)
    half = bytearray(t1); half[:2600] = rng.integers(97, 123, 2600, dtype=np.uint8).tobytes(); half = bytes(half)   # delta > a fifth, still a saving
    zeros = bytes(6000)
    chunks = [t1, v1, half, words_text(4000, seed=22).tobytes(), zeros, zeros[:5000] + b"\x01" * 10, rng.integers(0, 256, 3000, dtype=np.uint8).tobytes()]
    base = np.array([-1, 0, 0, 0, -1, 4, 3], np.int64)     # near-duplicate, half-duplicate, unrelated base, highly compressible + useless base, random + unrelated
    data = np.frombuffer(b"".join(chunks), np.uint8)
    cuts = np.concatenate([[0], np.cumsum([len(c) for c in chunks])]).astype(np.uint64)
    kinds = set()
    for pct in (0, 20, 60):
        cfg = orc.default_cfg(delta_max_ratio_pct=pct)
        out, off, kind = orc.deflate_chunks(data, cuts, cfg, None, base)
        for k, c in enumerate(chunks):
            b = chunks[base[k]] if base[k] >= 0 else None
            st, kd = pyref.deflate_record(c, b, delta_max_ratio_pct=pct)
            assert kd == int(kind[k]) and st == out[int(off[k]):int(off[k + 1])].tobytes(), (pct, k)
            assert _rt(st, b if kd == 2 else None) == c
            kinds.add((pct, k, kd))
    assert (0, 1, 2) in kinds and (0, 2, 2) in kinds and (20, 2, 0) in kinds and (0, 3, 0) in kinds and (0, 6, 0) in kinds
    # the quick accept really is taken without a look at FULL: a delta under a fifth of the chunk that does NOT beat FULL + 8
    st, kd = pyref.deflate_record(zeros[:5000] + b"\x01" * 10, zeros)
    full = pyref.deflate(zeros[:5000] + b"\x01" * 10)
    assert kd == 2 and len(st) + 8 >= len(full)


# ---------------------------------------------------------------- committed fixtures
def test_golden_pipeline_fixtures(orc):
    from hmse_amd import corpus
    data = corpus.wiki_synth(3 << 20, seed=42)
    assert h(data[: 1 << 20]) == FIX["wiki_synth_seed42_first_MiB_sha256"]
    for name, kw in (("default", {}), ("seed1", {"seed_base": 1}), ("reference_1_4_16", {"min_size": 1024, "avg_size": 4096, "max_size": 16384})):
        cfg = orc.default_cfg(seg_size=1 << 20, **kw)
        f = FIX[f"pipeline_3MiB_{name}"]
        cuts = orc.cdc(data, cfg)
        assert len(cuts) - 1 == f["n_chunks"] and [int(c) for c in cuts[:8]] == f["cuts_first8"] and h(cuts) == f["cuts_sha256"]
        dg = orc.sha256_chunks(data, cuts)
        assert h(dg) == f["digests_sha256"]
        assert h(orc.dedup(dg)[0]) == f["first_occ_sha256"]
        ids = np.arange(min(len(cuts) - 1, 64), dtype=np.uint64)
        sig = orc.minhash_chunks(data, cuts, cfg, ids)
        assert h(sig) == f["sig64_sha256"] and [int(v) for v in sig[0, :4]] == f["sig0_first4"]
        keys, base = orc.lsh(sig, cfg)
        assert h(keys) == f["band_keys64_sha256"] and base.tolist() == f["base64"]
        out, off, kind = orc.deflate_chunks(data, cuts, cfg, ids, base)
        assert int(off[-1]) == f["deflate64_total"] and h(out) == f["deflate64_sha256"]
    rnd = corpus.random_bytes(300_000)
    assert h(rnd) == FIX["random_0xDEADBEEF_300k"]["sha256"]
    assert h(orc.cdc(rnd, orc.default_cfg())) == FIX["random_0xDEADBEEF_300k"]["cuts_sha256"]


def test_golden_variants_family(orc):
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from make_golden import variants_dataset
    from hmse_amd import corpus
    var = variants_dataset(corpus.wiki_synth(3 << 20, seed=42))
    assert h(var) == FIX["variants_sha256"]
    f = FIX["pipeline_variants"]
    cfg = orc.default_cfg(seg_size=1 << 20)
    cuts = orc.cdc(var, cfg)
    fo, _ = orc.dedup(orc.sha256_chunks(var, cuts))
    uniq = np.nonzero(fo == np.arange(len(fo)))[0].astype(np.uint64)
    assert (len(cuts) - 1, len(uniq)) == (f["n_chunks"], f["n_unique"]) and h(cuts) == f["cuts_sha256"]
    sig = orc.minhash_chunks(var, cuts, cfg, uniq)
    _, base = orc.lsh(sig, cfg)
    out, off, kind = orc.deflate_chunks(var, cuts, cfg, uniq, base)
    assert base.tolist() == f["base"] and kind.tolist() == f["kind"] and np.diff(off).tolist() == f["stream_len"]
    assert h(out) == f["streams_sha256"] and (kind == 2).sum() > 10
    for k in np.nonzero(kind == 2)[0][:10]:                          # delta records inflate with zdict = base chunk
        b = int(uniq[base[k]]); c = int(uniq[k])
        assert _rt(out[int(off[k]):int(off[k + 1])].tobytes(), var[int(cuts[b]):int(cuts[b + 1])].tobytes()) == \
            var[int(cuts[c]):int(cuts[c + 1])].tobytes()


# ---------------------------------------------------------------- read path: inflate vs stock zlib (README.md:2397-2400)
def _zlib_raw_inflate(stream, zd=None):
    try:
        d = zlib.decompressobj(-15, zdict=zd) if zd else zlib.decompressobj(-15)
        out = d.decompress(stream) + d.flush()
        return out if d.eof and not d.unused_data else None
    except zlib.error:
        return None


def _zlib_streams():
    """(payload, dictionary, raw stream) over every block type zlib can emit: stored, fixed, dynamic, multi-block."""
    rng = np.random.default_rng(0)
    payloads = [b"", b"a", b"hello hello hello hello", rng.integers(0, 256, 3000, dtype=np.uint8).tobytes(), b"abc" * 5000,
                rng.integers(97, 101, 20000, dtype=np.uint8).tobytes(), b"\0" * 32768, words_text(32768, seed=3).tobytes()]
    zdict = words_text(40000, seed=3).tobytes()  # longer than the 32 KiB window: only its tail may be referenced
    for t in payloads:
        for lvl in (0, 1, 6, 9):
            for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
                for zd in (None, zdict):
                    c = zlib.compressobj(lvl, zlib.DEFLATED, -15, 9, strat, zdict=zd) if zd else zlib.compressobj(lvl, zlib.DEFLATED, -15, 9, strat)
                    yield t, zd, c.compress(t[:len(t) // 2]) + c.flush(zlib.Z_FULL_FLUSH) + c.compress(t[len(t) // 2:]) + c.flush()


def mutate(stream: bytes, rnd) -> bytes:
    m = bytearray(stream)
    k = rnd.random()
    if k < 0.5 and m:
        m[rnd.randrange(min(len(m), 40))] ^= 1 << rnd.randrange(8)   # headers and code-length sets live up front
    elif k < 0.7 and m:
        m[rnd.randrange(len(m))] ^= 1 << rnd.randrange(8)
    elif k < 0.85:
        m = m[:rnd.randrange(len(m) + 1)]
    else:
        m += b"\0"
    return bytes(m)


def test_inflate_equals_zlib_on_valid_streams(orc):
    n = 0
    for t, zd, s in _zlib_streams():
        got, rc = orc.inflate(s, len(t), zd)
        assert rc == 0 and got == t, (len(t), rc)
        n += 1
    assert n == 8 * 4 * 4 * 2


def test_inflate_rejects_exactly_what_zlib_rejects(orc):
    """Mutated streams: same accept/reject decision as stock zlib 1.2.11 (+ the raw-length contract), same bytes if accepted."""
    import random
    rnd = random.Random(1)
    accepted = rejected = 0
    for t, zd, s in _zlib_streams():
        for _ in range(12):
            m = mutate(s, rnd)
            want = _zlib_raw_inflate(m, zd)
            if want is not None and len(want) != len(t):
                want = None
            got, rc = orc.inflate(m, len(t), zd)
            assert (got is None) == (want is None), (len(t), rc)
            assert got == want
            accepted += got is not None
            rejected += got is None
    assert accepted > 50 and rejected > 1000


def test_inflate_chunks_inverts_deflate_chunks(orc, corpus_small):
    cfg = orc.default_cfg()
    data = corpus_small[: 600_000]
    cuts = orc.cdc(data, cfg)
    n = len(cuts) - 1
    base = np.full(n, -1, dtype=np.int64)
    base[3::2] = np.arange(3, n, 2) - 2
    out, off, kind = orc.deflate_chunks(data, cuts, cfg, None, base)
    raw, raw_off, ok = orc.inflate_chunks(out, off, kind, base, np.diff(cuts.astype(np.int64)))
    assert ok.all() and np.array_equal(raw_off, cuts) and np.array_equal(raw, data[: int(cuts[-1])])
    # a corrupt record poisons the DELTA records that (transitively) use it as dictionary, nothing else
    k = int(np.nonzero(kind == 2)[0][4]); b = int(base[k])
    bad = out.copy(); bad[int(off[b])] ^= 0x06  # block type bits of the base's stream -> reserved type 3
    _, _, ok2 = orc.inflate_chunks(bad, off, kind, base, np.diff(cuts.astype(np.int64)))
    poisoned = {b}
    for j in range(n):
        if kind[j] == 2 and int(base[j]) in poisoned:
            poisoned.add(j)
    assert k in poisoned and set(np.nonzero(~ok2)[0].tolist()) == poisoned
