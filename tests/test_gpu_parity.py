"""GPU parity: the HIP path (through the C-ABI) vs the CPU oracle on the same seeded inputs.

Bar: bit-exact (integer/byte work).  Sizes are what the oracle finishes in seconds; full-size
properties live in test_gpu_properties.py.
"""
import hashlib

import numpy as np
import pytest

from conftest import words_text

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def to_dev(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def ocfg(orc, cfg):
    from dataclasses import asdict
    return orc.default_cfg(**asdict(cfg))


CASES = [
    ("empty", lambda: np.zeros(0, np.uint8)),
    ("one", lambda: np.array([7], np.uint8)),
    ("63", lambda: words_text(63)),
    ("min-1", lambda: words_text(2047)),
    ("min", lambda: words_text(2048)),
    ("tile", lambda: words_text(32768)),
    ("tile+1", lambda: words_text(32769)),
    ("ragged", lambda: words_text(1_000_003, seed=7)),
    ("zeros", lambda: np.zeros(300_000, np.uint8)),
    ("ones", lambda: np.full(300_000, 255, np.uint8)),
    ("random", lambda: np.random.Generator(np.random.PCG64(0xDEADBEEF)).integers(0, 256, 1_500_000, dtype=np.uint8)),
    ("period64", lambda: np.tile(np.arange(64, dtype=np.uint8), 8192)),
]


@pytest.mark.parametrize("name,gen", CASES, ids=[c[0] for c in CASES])
def test_l2_cuts_bit_exact(name, gen, orc, dev):
    from hmse_amd import IngestConfig, ops
    data = gen()
    for cfg in (IngestConfig(), IngestConfig.reference_preset(), IngestConfig(seg_size=100_000)):
        want = orc.cdc(data, ocfg(orc, cfg))
        got = ops.l2_cdc(to_dev(data, dev), cfg).cpu().numpy().astype(np.uint64)
        assert got.shape == want.shape, (name, cfg, got.shape, want.shape)
        assert np.array_equal(got, want), (name, cfg)


def test_l2_corpus_and_custom_segments(orc, dev, corpus_small):
    import torch
    from hmse_amd import IngestConfig, ops
    cfg = IngestConfig()
    d = to_dev(corpus_small, dev)
    want = orc.cdc(corpus_small, ocfg(orc, cfg))
    got = ops.l2_cdc(d, cfg).cpu().numpy().astype(np.uint64)
    assert np.array_equal(got, want)
    # document-style ragged segments, including an empty one
    n = corpus_small.size
    seg = np.array([0, 1, 1, 70_000, 70_001, 3_000_000, n], dtype=np.uint64)
    want = orc.cdc(corpus_small, ocfg(orc, cfg), seg)
    got = ops.l2_cdc(d, cfg, torch.from_numpy(seg.astype(np.int64)).to(dev)).cpu().numpy().astype(np.uint64)
    assert np.array_equal(got, want)


def test_l3_sha256_bit_exact(orc, dev, corpus_small):
    from hmse_amd import IngestConfig, ops
    cfg = IngestConfig()
    cuts = orc.cdc(corpus_small, ocfg(orc, cfg))
    got = ops.l3_sha256(to_dev(corpus_small, dev), to_dev(cuts.astype(np.int64), dev)).cpu().numpy()
    for i in range(len(cuts) - 1):
        assert got[i].tobytes() == hashlib.sha256(corpus_small[int(cuts[i]):int(cuts[i + 1])].tobytes()).digest(), i


def test_l3_sha256_padding_edges(dev):
    """Every length 0..200 (all padding branches) and chunks ending exactly at the buffer end."""
    from hmse_amd import ops
    rng = np.random.default_rng(1)
    lens = list(range(0, 201)) + [4095, 4096, 4097, 32768]
    data = rng.integers(0, 256, sum(lens), dtype=np.uint8)
    cuts = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    got = ops.l3_sha256(to_dev(data, dev), to_dev(cuts, dev)).cpu().numpy()
    for i, L in enumerate(lens):
        assert got[i].tobytes() == hashlib.sha256(data[cuts[i]:cuts[i + 1]].tobytes()).digest(), L


def test_l3_dedup_first_occurrence(orc, dev):
    from hmse_amd import ops
    rng = np.random.default_rng(3)
    uniq = rng.integers(0, 256, (5000, 32), dtype=np.uint8)
    pick = rng.integers(0, 5000, 40000)
    pick[:100] = 17  # one heavy duplicate class
    dg = uniq[pick]
    fo_w, rc_w = orc.dedup(dg)
    fo, rc = ops.l3_dedup(to_dev(dg, dev))
    assert np.array_equal(fo.cpu().numpy().astype(np.uint64), fo_w)
    assert np.array_equal(rc.cpu().numpy().astype(np.uint32), rc_w)


def test_l4_minhash_bit_exact(orc, dev, corpus_small):
    from hmse_amd import IngestConfig, ops
    for cfg in (IngestConfig(), IngestConfig(seed_base=1)):
        data = corpus_small[: 600_000]
        cuts = orc.cdc(data, ocfg(orc, cfg))
        want = orc.minhash_chunks(data, cuts, ocfg(orc, cfg))
        got = ops.l4_minhash(to_dev(data, dev), to_dev(cuts.astype(np.int64), dev), cfg).cpu().numpy().view(np.uint32)
        assert np.array_equal(got, want)


def test_l4_minhash_edges_and_selection(orc, dev):
    from hmse_amd import IngestConfig, ops
    cfg = IngestConfig()
    rng = np.random.default_rng(5)
    lens = [0, 1, 3, 4, 5, 64, 8191, 8195, 20000, 32768]
    data = rng.integers(0, 256, sum(lens), dtype=np.uint8)
    data[-32768:] = np.tile(np.frombuffer(b"abcd", np.uint8), 8192)  # one distinct-shingle-poor chunk
    cuts = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    ids = np.array([9, 0, 3, 8, 2, 7], dtype=np.uint64)
    want = orc.minhash_chunks(data, cuts, ocfg(orc, cfg), ids)
    got = ops.l4_minhash(to_dev(data, dev), to_dev(cuts.astype(np.int64), dev), cfg, to_dev(ids.astype(np.int64), dev))
    assert np.array_equal(got.cpu().numpy().view(np.uint32), want)
    assert (want[1] == 0xFFFFFFFF).all() and (want[4] == 0xFFFFFFFF).all()  # len < 4 (README.md:2585 underflow guard)


def test_l4_minhash_memo_table_changes_nothing(orc, dev):
    """The memo table of hmse_l4_minhash (which seeds of a shingle hash below 2^22, looked up instead of recomputed) is an
    optimisation only: with and without it the signatures are identical — on text (warm table: launches of growing size, ~7 k
    chunks), on random bytes (no repeating 4-grams: the table fills up, wavefronts stop looking), on tiny chunks (most
    seeds end above the threshold and are re-evaluated), on chunks above 12 KiB (two passes) — and equal the oracle's."""
    import torch
    from hmse_amd import IngestConfig, corpus, ops
    cfg = IngestConfig()
    rng = np.random.default_rng(3)
    text = corpus.wiki_synth(64 << 20, seed=42)
    for name, data in (("text", text), ("random", rng.integers(0, 256, 24 << 20, dtype=np.uint8))):
        d = to_dev(data, dev)
        cuts = ops.l2_cdc(d, cfg)
        a = ops.l4_minhash(d, cuts, cfg)
        b = ops.l4_minhash(d, cuts, cfg, memo=False)
        assert torch.equal(a, b), name
        k = 40
        want = orc.minhash_chunks(data, cuts.cpu().numpy().astype(np.uint64), ocfg(orc, cfg), np.arange(cuts.numel() - 1 - k, cuts.numel() - 1, dtype=np.uint64))
        assert np.array_equal(a[-k:].cpu().numpy().view(np.uint32), want), name      # the last chunks: the warmest table
    # fixed tiny / large chunks over the same text
    for size in (64, 300, 2048, 20000, 32768):
        n = (8 << 20) // size * size
        cuts = torch.arange(0, n + 1, size, dtype=torch.int64, device=dev)
        d = to_dev(text[:n], dev)
        assert torch.equal(ops.l4_minhash(d, cuts, cfg), ops.l4_minhash(d, cuts, cfg, memo=False)), size


def test_l1_deflate_in_pieces_is_the_same_deflate(dev):
    """ops.l1_deflate bounds the C-ABI call's record workspace: a selection whose per-job records exceed `ws_limit` is encoded
    in consecutive pieces (dictionaries named by chunk id, so a piece may use a chunk of an earlier piece) — same streams,
    offsets and kinds as the single call, with and without a selection and dictionaries."""
    import torch
    from hmse_amd import IngestConfig, corpus, ops
    cfg = IngestConfig()
    d = to_dev(corpus.wiki_synth(24 << 20, seed=42), dev)
    cuts = ops.l2_cdc(d, cfg)
    n = cuts.numel() - 1
    ids = torch.arange(3, n, 2, dtype=torch.int64, device=dev)
    sig = ops.l4_minhash(d, cuts, cfg, ids)
    _, base = ops.l4_lsh(sig, cfg)
    assert int((base >= 0).sum()) > 20
    for sel, b in ((None, None), (ids, None), (ids, base)):
        one = ops.l1_deflate(d, cuts, cfg, sel, b)
        many = ops.l1_deflate(d, cuts, cfg, sel, b, ws_limit=12 << 20)            # ~6-12 pieces
        for x, y in zip(one, many):
            assert torch.equal(x, y)


def test_l4_lsh_bit_exact(orc, dev):
    from hmse_amd import IngestConfig, ops
    cfg = IngestConfig()
    rng = np.random.default_rng(9)
    n = 3000
    sig = rng.integers(0, 2**32, (n, 128), dtype=np.uint64).astype(np.uint32)
    for i in range(200, n):  # plant band matches: copy one band of an earlier signature
        if rng.random() < 0.3:
            j = int(rng.integers(0, i)); b = int(rng.integers(0, 4))
            sig[i, 32 * b:32 * b + 32] = sig[j, 32 * b:32 * b + 32]
    sig[5] = 0xFFFFFFFF; sig[77] = 0xFFFFFFFF
    keys_w, base_w = orc.lsh(sig, ocfg(orc, cfg))
    keys, base = ops.l4_lsh(to_dev(sig.view(np.int32), dev), cfg)
    assert np.array_equal(keys.cpu().numpy().view(np.uint32), keys_w)
    assert np.array_equal(base.cpu().numpy(), base_w)
    assert (base_w >= 0).sum() > 500


def _roundtrip(stream: bytes, zdict: bytes | None = None) -> bytes:
    import zlib
    d = zlib.decompressobj(-15, zdict=zdict) if zdict else zlib.decompressobj(-15)
    out = d.decompress(stream) + d.flush()
    assert d.eof and not d.unused_data
    return out


def test_l1_deflate_bit_exact_and_roundtrip(orc, dev, corpus_small):
    """Streams equal the oracle's byte for byte, inflate through stock zlib, FULL/DELTA kinds agree."""
    from hmse_amd import IngestConfig, ops
    cfg = IngestConfig()
    data = corpus_small[: 1_500_000]
    cuts = orc.cdc(data, ocfg(orc, cfg))
    n = len(cuts) - 1
    base = np.full(n, -1, dtype=np.int64)
    base[5::3] = np.arange(5, n, 3) - 4          # arbitrary earlier chunks as dictionaries
    want_out, want_off, want_kind = orc.deflate_chunks(data, cuts, ocfg(orc, cfg), None, base)
    out, off, kind = ops.l1_deflate(to_dev(data, dev), to_dev(cuts.astype(np.int64), dev), cfg, None, to_dev(base, dev))
    out, off, kind = out.cpu().numpy(), off.cpu().numpy().astype(np.uint64), kind.cpu().numpy()
    assert np.array_equal(off, want_off)
    assert np.array_equal(kind, want_kind)
    assert np.array_equal(out, want_out)
    for k in range(n):
        chunk = data[int(cuts[k]):int(cuts[k + 1])].tobytes()
        zd = data[int(cuts[base[k]]):int(cuts[base[k] + 1])].tobytes() if kind[k] == 2 else None
        assert _roundtrip(out[int(off[k]):int(off[k + 1])].tobytes(), zd) == chunk, k


def test_l1_deflate_edges(orc, dev):
    """Tiny chunks, incompressible (stored), one-symbol runs (huge bucket, 258-matches), near-duplicates with dict."""
    from hmse_amd import IngestConfig, ops
    cfg = IngestConfig()
    rng = np.random.Generator(np.random.PCG64(0xDEADBEEF))
    text = words_text(20000, seed=11)
    variant = text.copy(); variant[rng.integers(0, 20000, 40)] = 63
    parts = [np.array([65], np.uint8), np.frombuffer(b"ab", np.uint8), np.frombuffer(b"abc", np.uint8),
             np.frombuffer(b"abcd", np.uint8), np.frombuffer(b"abcabcabcabc", np.uint8),
             rng.integers(0, 256, 5000, dtype=np.uint8), np.zeros(20000, np.uint8), np.full(32768, 0x61, np.uint8),
             text, variant, np.tile(np.arange(256, dtype=np.uint8), 64), rng.integers(0, 4, 9000, dtype=np.uint8)]
    data = np.concatenate(parts)
    cuts = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.uint64)
    base = np.full(len(parts), -1, dtype=np.int64)
    base[9] = 8     # near-duplicate with its original as dictionary
    base[5] = 4     # useless dictionary: must stay FULL
    base[7] = 6
    for lvl_cfg in (cfg, IngestConfig(level=1), IngestConfig(chain_depth=200)):
        want_out, want_off, want_kind = orc.deflate_chunks(data, cuts, ocfg(orc, lvl_cfg), None, base)
        out, off, kind = ops.l1_deflate(to_dev(data, dev), to_dev(cuts.astype(np.int64), dev), lvl_cfg, None, to_dev(base, dev))
        assert np.array_equal(off.cpu().numpy().astype(np.uint64), want_off)
        assert np.array_equal(kind.cpu().numpy(), want_kind)
        assert np.array_equal(out.cpu().numpy(), want_out)
    assert want_kind[9] == 2 and want_kind[5] == 0


def test_l1_deflate_dictionary_jobs_adversarial(orc, dev):
    """Dictionary jobs of every size class under inputs that stress the round-3 paths (rule 2c hint pass, the wavefront's window
    queue, rule 7's second pass): identical chunk and dictionary (every position hinted: the queue finds no work), unrelated
    dictionary (nothing hinted: every position walks, the delta is refused and FULL comes from the second pass), one hot bucket
    (a 4-byte period: thousands of candidates per position), runs that end inside the first 16 bytes, dictionaries longer than
    the window, 32 KiB + 32 KiB (class B), tiny chunks with tiny dictionaries, edits at block boundaries."""
    from hmse_amd import IngestConfig, ops
    rng = np.random.Generator(np.random.PCG64(4242))
    text = words_text(70000, seed=23)
    def edit(a, k, seed):
        r = np.random.default_rng(seed); v = a.copy(); v[r.integers(0, len(v), k)] = 35; return v
    period = np.tile(np.frombuffer(b"abcd", np.uint8), 6000)
    parts, base = [], []
    def add(chunk, dict_=None):
        if dict_ is not None:
            parts.append(np.asarray(dict_, np.uint8)); base.append(-1)
            parts.append(np.asarray(chunk, np.uint8)); base.append(len(parts) - 2)
        else:
            parts.append(np.asarray(chunk, np.uint8)); base.append(-1)
    for L in (2048, 4500, 6100, 8000, 10700, 16000, 32768):                 # T = 2L: classes S .. B
        a = text[100:100 + L]
        add(a.copy(), a)                                                   # identical
        add(edit(a, 3, L), a)                                              # near-duplicate
        add(rng.integers(0, 256, L, dtype=np.uint8), a)                    # unrelated dictionary
        v = a.copy(); v[63::64] ^= 1; add(v, a)                            # an edit in every 64-byte block: anchors fail or runs are short
    add(period[:9000], period[:7000]); add(period[:20000], period[2:20002])   # one hot bucket, shifted diagonal
    add(np.zeros(30000, np.uint8), np.zeros(32768, np.uint8))
    add(text[:5], text[:7]); add(text[:3], text[:2]); add(text[:70], text[3:60])   # tiny
    add(np.concatenate([text[:3000], text[:3000]]), edit(text[:3000], 2, 1))       # a repeat inside the chunk AND a dictionary (rule 2c takes the hint)
    data = np.concatenate(parts)
    cuts = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.uint64)
    base = np.array(base, np.int64)
    for lvl_cfg in (IngestConfig(), IngestConfig(chain_depth=3), IngestConfig(delta_max_ratio_pct=20)):
        want_out, want_off, want_kind = orc.deflate_chunks(data, cuts, ocfg(orc, lvl_cfg), None, base)
        out, off, kind = ops.l1_deflate(to_dev(data, dev), to_dev(cuts.astype(np.int64), dev), lvl_cfg, None, to_dev(base, dev))
        assert np.array_equal(kind.cpu().numpy(), want_kind)
        assert np.array_equal(off.cpu().numpy().astype(np.uint64), want_off)
        assert np.array_equal(out.cpu().numpy(), want_out)
    k = want_kind[np.array(base) >= 0]
    assert (k == 2).sum() >= 15 and (k == 0).sum() >= 7                   # both outcomes of rule 7 occur (incl. FULL via the second pass)
    import zlib
    o = out.cpu().numpy()
    for j in range(len(parts)):                                            # every record inflates through stock zlib
        st = o[int(want_off[j]):int(want_off[j + 1])].tobytes()
        d = zlib.decompressobj(-15, zdict=parts[base[j]].tobytes()[-32768:]) if want_kind[j] == 2 else zlib.decompressobj(-15)
        assert d.decompress(st) == parts[j].tobytes()


@pytest.mark.parametrize("L,n_copies", [(4096, 1600), (10000, 1100)], ids=["class-S", "class-SG2"])
def test_l1_deflate_more_dictionary_jobs_of_one_class_than_resident_workgroups(L, n_copies, orc, dev):
    """the round-3 hang of the dictionary matcher's work queue needed MORE dictionary jobs of one size class
    than the launch has workgroups (grid = min(2 * jobs, 512)): a workgroup then takes a second, third ... dictionary job, the
    next job's metadata is handed over in LDS (`sm.nx`) and the wavefront queue restarts.  One base chunk, `n_copies` lightly edited
    copies that all name it as their dictionary — window 2 L: class S (L = 4096) and class SG2 (L = 10000) — streams, offsets and
    kinds equal the oracle's; ops.l1_deflate raises on any device status bit (bit 4 = the trip budget), so status is 0."""
    from hmse_amd import IngestConfig, ops
    cfg = IngestConfig()
    rng = np.random.default_rng(L)
    base_chunk = words_text(L, seed=L)
    parts = [base_chunk]
    for i in range(n_copies):
        v = base_chunk.copy()
        k = 1 + i % 7
        v[rng.integers(0, L, k)] = rng.integers(32, 127, k, dtype=np.uint8)
        parts.append(v)
    data = np.concatenate(parts)
    cuts = (np.arange(len(parts) + 1, dtype=np.uint64) * np.uint64(L))
    base = np.zeros(len(parts), dtype=np.int64); base[0] = -1
    want_out, want_off, want_kind = orc.deflate_chunks(data, cuts, ocfg(orc, cfg), None, base)
    out, off, kind = ops.l1_deflate(to_dev(data, dev), to_dev(cuts.astype(np.int64), dev), cfg, None, to_dev(base, dev))
    assert np.array_equal(kind.cpu().numpy(), want_kind)
    assert np.array_equal(off.cpu().numpy().astype(np.uint64), want_off)
    assert np.array_equal(out.cpu().numpy(), want_out)
    assert (want_kind[1:] == 2).all()


def test_l1_deflate_near_incompressible_record_slots(orc, dev):
    """Streams whose size is within a byte of the stored size (L + 4 of L + 5), at 256 consecutive chunk lengths: the
    encode kernel copies its bit image out in whole 16-byte stores, which must not reach the neighbouring job record
    (its literal histogram) whatever the record padding is — every stream must still equal the oracle's."""
    from hmse_amd import IngestConfig, ops
    cfg = IngestConfig()
    oc = ocfg(orc, cfg)
    rng = np.random.Generator(np.random.PCG64(1234))
    parts = []
    for L in range(1500, 1756):
        body = rng.integers(0, 256, L, dtype=np.uint8)
        for t in range(16, 96):   # a repeated tail just long enough for the dynamic block to beat the stored one
            c = body.copy(); c[L - t:] = c[:t]
            out, off, _ = orc.deflate_chunks(c, np.array([0, L], dtype=np.uint64), oc, None, None)
            if ((out[0] >> 1) & 3) != 0 and int(off[1]) <= L + 4:
                break
        else:
            raise AssertionError("no near-incompressible variant found")
        assert L - 8 < int(off[1]) <= L + 4
        parts.append(c)
    order = rng.permutation(len(parts))
    parts = [parts[i] for i in order]
    data = np.concatenate(parts)
    cuts = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.uint64)
    base = np.full(len(parts), -1, dtype=np.int64)
    base[3::4] = np.arange(3, len(parts), 4) - 2   # some jobs also carry a (useless) dictionary: FULL + DELTA records side by side
    want_out, want_off, want_kind = orc.deflate_chunks(data, cuts, oc, None, base)
    d_data, d_cuts, d_base = to_dev(data, dev), to_dev(cuts.astype(np.int64), dev), to_dev(base, dev)
    for _ in range(3):   # the jobs run concurrently: an overrun would hit a neighbour only some of the time
        out, off, kind = ops.l1_deflate(d_data, d_cuts, cfg, None, d_base)
        assert np.array_equal(off.cpu().numpy().astype(np.uint64), want_off)
        assert np.array_equal(kind.cpu().numpy(), want_kind)
        assert np.array_equal(out.cpu().numpy(), want_out)


def test_l1_encode_window_slides_over_long_streams(orc, dev):
    """The encode kernel assembles a record's bit image in a 4 KiB LDS window that slides over the stream (round 4): records whose
    stream is several windows long in both record classes (chunks <= 12 KiB and above) — 5..7 bits of entropy per byte, so the
    dynamic block beats the stored one and the stream is 0.6..0.9 of the chunk —, streams that end within a few bytes of a window
    edge, and text in between.  Streams equal the oracle's and inflate through stock zlib."""
    import zlib
    from hmse_amd import IngestConfig, ops
    cfg = IngestConfig()
    oc = ocfg(orc, cfg)
    rng = np.random.Generator(np.random.PCG64(77))
    text = words_text(40000, seed=5)
    parts = []
    for L, hi in ((12288, 64), (12288, 128), (12000, 32), (9000, 200), (8192, 16), (32768, 128), (30000, 64), (20000, 32), (16384, 250)):
        parts.append(rng.integers(0, hi, L, dtype=np.uint8))
        parts.append(text[: 3000 + 7 * len(parts)])
    # stream lengths around the window edge: 5-bit symbols -> ~0.63 bytes of stream per byte; lengths stepped so that the ends of the
    # streams sweep across 4096 and 8192 bytes
    for L in list(range(6450, 6650, 8)) + list(range(12950, 13150, 8)):
        parts.append(rng.integers(0, 32, L, dtype=np.uint8))
    data = np.concatenate(parts)
    cuts = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.uint64)
    want_out, want_off, want_kind = orc.deflate_chunks(data, cuts, oc, None, None)
    sizes = np.diff(want_off.astype(np.int64))
    assert (sizes > 4096).sum() >= 10 and (sizes > 8192).sum() >= 5 and (np.abs(sizes - 4096) < 24).any() and (np.abs(sizes - 8192) < 24).any()
    out, off, kind = ops.l1_deflate(to_dev(data, dev), to_dev(cuts.astype(np.int64), dev), cfg, None, None)
    assert np.array_equal(off.cpu().numpy().astype(np.uint64), want_off)
    assert np.array_equal(kind.cpu().numpy(), want_kind)
    o = out.cpu().numpy()
    assert np.array_equal(o, want_out)
    for j in range(len(parts)):
        assert zlib.decompressobj(-15).decompress(o[int(want_off[j]):int(want_off[j + 1])].tobytes()) == parts[j].tobytes()


def test_l1_encode_spill_path_gives_the_same_streams(dev):
    """A FULL record's stream is written over its token list window by window; a record for which that could overrun tokens still to
    be read (more than 32 bits per token on average: never seen) parks its finished windows in the global scratch.  The path is forced
    for every record in a child process (HMSE_ENC_FORCE_SPILL is read once per process): same streams as the oracle."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HMSE_ENC_FORCE_SPILL="1")
    r = subprocess.run([sys.executable, os.path.join(here, "enc_spill_check.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "spill path OK" in r.stdout


def test_l1_deflate_selection(orc, dev, corpus_small):
    from hmse_amd import IngestConfig, ops
    cfg = IngestConfig()
    data = corpus_small[: 400_000]
    cuts = orc.cdc(data, ocfg(orc, cfg))
    n = len(cuts) - 1
    ids = np.array([n - 1, 0, 7, 3, 12, 5], dtype=np.uint64)
    base = np.array([-1, -1, 1, 2, 0, -1], dtype=np.int64)   # indices into the selection
    want_out, want_off, want_kind = orc.deflate_chunks(data, cuts, ocfg(orc, cfg), ids, base)
    out, off, kind = ops.l1_deflate(to_dev(data, dev), to_dev(cuts.astype(np.int64), dev), cfg, to_dev(ids.astype(np.int64), dev), to_dev(base, dev))
    assert np.array_equal(off.cpu().numpy().astype(np.uint64), want_off)
    assert np.array_equal(out.cpu().numpy(), want_out)
    assert np.array_equal(kind.cpu().numpy(), want_kind)


def test_torch_ops_hmse_equal_the_operator_functions(dev, corpus_small):
    """torch.ops.hmse.*  run the same C-ABI calls as hmse_amd.ops."""
    import torch
    from hmse_amd import IngestConfig, ops, torch_ops  # noqa: F401
    cfg = IngestConfig()
    d = to_dev(corpus_small[: 2 << 20], dev)
    cuts = torch.ops.hmse.l2_cdc(d)
    assert torch.equal(cuts, ops.l2_cdc(d, cfg))
    dg = torch.ops.hmse.l3_sha256(d, cuts)
    assert torch.equal(dg, ops.l3_sha256(d, cuts))
    fo, rc = torch.ops.hmse.l3_dedup(dg)
    uniq = (fo == torch.arange(fo.numel(), device=dev)).nonzero().flatten()
    sig = torch.ops.hmse.l4_minhash(d, cuts, uniq)
    assert torch.equal(sig, ops.l4_minhash(d, cuts, cfg, uniq))
    keys, base = torch.ops.hmse.l4_lsh(sig)
    out, off, kind = torch.ops.hmse.l1_deflate(d, cuts, uniq, base)
    o2, f2, k2 = ops.l1_deflate(d, cuts, cfg, uniq, base)
    assert torch.equal(out, o2) and torch.equal(off, f2) and torch.equal(kind, k2)
    lens = (cuts[1:] - cuts[:-1])[uniq]
    raw, raw_off, ok = torch.ops.hmse.l1_inflate(out, off, kind, base, lens)
    assert bool(ok.all()) and raw.numel() == int(lens.sum())
