"""GPU: the read path — hmse_l1_inflate / hmse_read_assemble through the C-ABI against the CPU oracle
(itself pinned against stock zlib in test_oracle.py).  Bar: byte-exact output and the same per-record accept/reject
decision; then whole-shard and manifest round trips verified by the L3 SHA-256 kernel (README.md:1329, 1621-1675)."""
import random

import numpy as np
import pytest

from conftest import words_text
from test_oracle import _zlib_streams, mutate

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@pytest.fixture(autouse=True, params=[1, 2], ids=["wave-per-stream", "lane-per-stream"])
def decoder(request, dev):
    """Every test here runs under both decoders of hmse_l1_inflate (include/hmse.h hmse_l1_inflate_mode)."""
    from hmse_amd import ops
    ops.l1_inflate_mode(request.param)
    yield request.param
    ops.l1_inflate_mode(0)


def to_dev(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def pack(records):
    """[(stream bytes, raw_len, dictionary slot or -1)] -> dense arrays of the hmse_l1_inflate call."""
    streams = np.frombuffer(b"".join(r[0] for r in records) + b"\0", np.uint8)[:-1].copy()
    off = np.concatenate([[0], np.cumsum([len(r[0]) for r in records])]).astype(np.int64)
    raw_len = np.array([r[1] for r in records], np.int64)
    base = np.array([r[2] for r in records], np.int64)
    kind = np.where(base >= 0, 2, 0).astype(np.uint8)
    return streams, off, kind, base, raw_len


def run_both(orc, dev, records):
    from hmse_amd import ops
    streams, off, kind, base, raw_len = pack(records)
    want_raw, want_off, want_ok = orc.inflate_chunks(streams, off, kind, base, raw_len)
    s = to_dev(streams if streams.size else np.zeros(1, np.uint8), dev)
    raw, raw_off, ok = ops.l1_inflate(s[: streams.size] if streams.size else s, to_dev(off, dev), to_dev(kind, dev), to_dev(base, dev),
                                      to_dev(raw_len, dev), check=False)
    raw, raw_off, ok = raw.cpu().numpy(), raw_off.cpu().numpy().astype(np.uint64), ok.cpu().numpy().astype(bool)
    assert np.array_equal(raw_off, want_off)
    assert np.array_equal(ok, want_ok), np.nonzero(ok != want_ok)[0][:10]
    for k in np.nonzero(want_ok)[0]:
        a, b = int(want_off[k]), int(want_off[k + 1])
        assert np.array_equal(raw[a:b], want_raw[a:b]), k
    return want_ok


def zlib_records():
    """Every block type stock zlib emits (stored / fixed / dynamic / multi-block), each also with a preset dictionary:
    the dictionary is record 0 itself (stored), as a DeltaChunk's base is in the manifest."""
    import zlib
    zdict = None
    recs = []
    for t, zd, s in _zlib_streams():
        if zd is not None and zdict is None:
            zdict = zd
            c = zlib.compressobj(0, zlib.DEFLATED, -15)
            recs.insert(0, (c.compress(zd) + c.flush(), len(zd), -1))
    for t, zd, s in _zlib_streams():
        recs.append((s, len(t), 0 if zd is not None else -1))
    return recs


def test_inflate_zlib_streams_bit_exact(orc, dev):
    recs = zlib_records()
    ok = run_both(orc, dev, recs)
    assert ok.all() and len(recs) == 257


def test_inflate_accepts_and_rejects_like_the_oracle(orc, dev):
    """Mutated records (bit flips in headers/code sets/payload, truncation, trailing bytes, wrong raw length): the device
    flags exactly the records the oracle (== stock zlib + the length contract) rejects, decodes all others, never hangs."""
    rnd = random.Random(7)
    recs = zlib_records()
    mutated = [recs[0]]
    for s, n, b in recs[1:]:
        for _ in range(6):
            mutated.append((mutate(s, rnd), n + (rnd.random() < 0.05), b))
    ok = run_both(orc, dev, mutated)
    assert 100 < ok.sum() < len(mutated) - 500
    # a corrupt dictionary record poisons its DELTA records and nothing else
    poisoned = [(b"\x07" + recs[0][0][1:], recs[0][1], -1)] + recs[1:]  # reserved block type 3
    ok = run_both(orc, dev, poisoned)
    assert not ok[0] and all(ok[i] == (poisoned[i][2] < 0) for i in range(1, len(poisoned)))


def test_inflate_edges(orc, dev):
    """Empty selection, empty chunks, self-overlapping matches of every small distance, 258-byte runs, maximum chunk
    with maximum dictionary, distances reaching the first dictionary byte, chained dictionaries."""
    import torch, zlib
    from hmse_amd import ops
    raw, raw_off, ok = ops.l1_inflate(torch.zeros(1, dtype=torch.uint8, device=dev), torch.zeros(1, dtype=torch.int64, device=dev),
                                      torch.zeros(0, dtype=torch.uint8, device=dev), None, torch.zeros(0, dtype=torch.int64, device=dev))
    assert raw.numel() == 0 and raw_off.tolist() == [0]

    def z(t, zd=None, lvl=9):
        c = zlib.compressobj(lvl, zlib.DEFLATED, -15, 9, zdict=zd) if zd else zlib.compressobj(lvl, zlib.DEFLATED, -15, 9)
        return c.compress(t) + c.flush()
    rng = np.random.default_rng(5)
    big = words_text(32768, seed=9).tobytes()
    big2 = bytearray(big); big2[100] ^= 1; big2 = bytes(big2)     # match of distance 32768 into the dictionary's first bytes
    recs = [(z(b""), 0, -1), (z(b"x"), 1, -1)]
    for d in list(range(1, 20)) + [31, 32, 33, 63, 64, 65, 127, 128, 129, 257, 258, 259]:
        period = rng.integers(0, 256, d, dtype=np.uint8).tobytes()
        recs.append((z(period * (3000 // d + 2)), len(period) * (3000 // d + 2), -1))
    recs.append((z(big), len(big), -1))
    i_big = len(recs) - 1
    recs.append((z(big2, big), len(big2), i_big))
    recs.append((z(big, big2), len(big), i_big + 1))                # chained: dictionary is itself a DELTA record
    recs.append((z(b""), 0, i_big))
    assert run_both(orc, dev, recs).all()


def test_inflate_inverts_gpu_deflate_with_dictionaries(orc, dev, corpus_small):
    from hmse_amd import IngestConfig, ops
    import torch
    cfg = IngestConfig()
    data = to_dev(corpus_small[: 2_000_000], dev)
    cuts = ops.l2_cdc(data, cfg)
    n = cuts.numel() - 1
    base = torch.full((n,), -1, dtype=torch.int64, device=dev)
    base[5::3] = torch.arange(5, n, 3, device=dev) - 4
    out, off, kind = ops.l1_deflate(data, cuts, cfg, None, base)
    raw, raw_off, ok = ops.l1_inflate(out, off, kind, base, cuts[1:] - cuts[:-1])
    assert bool(ok.all()) and torch.equal(raw_off, cuts) and torch.equal(raw, data[: int(cuts[-1])])
    assert int((kind == 2).sum()) > 10


def test_shard_and_manifest_round_trip_on_gpu(dev):
    """ingest -> (ShardResult | manifest bytes) -> GPU read path -> identical bytes, every chunk's SHA-256 re-verified."""
    import os, sys, torch
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import variants_dataset
    from hmse_amd import IngestConfig, corpus, ingest, manifest, read
    cfg = IngestConfig(seg_size=1 << 20)
    data = np.concatenate([variants_dataset(corpus.wiki_synth(3 << 20, seed=42)), corpus.wiki_synth(2 << 20, seed=42)])
    d = torch.from_numpy(data).to(dev)
    res = ingest.ingest_shard(d, cfg)
    assert res.stats["pointer"] > 0 and res.stats["delta"] > 0
    assert torch.equal(read.reconstruct_shard(res), d)
    m = manifest.Manifest.from_bytes(manifest.build_manifest(res).to_bytes())
    assert torch.equal(read.read_manifest(m, dev), d)
    assert manifest.reconstruct(m) == data.tobytes()          # and the host verifier (stock zlib) agrees
    # a flipped payload byte is caught: by the decoder, or by the SHA-256 gate behind it
    blob = m.blob.copy(); blob[int(m.index["lba"][3]) * m.lba_unit + 20] ^= 0x40
    m_bad = manifest.Manifest(m.lba_unit, m.index, m.chunk_map, m.pointers, blob)
    with pytest.raises((read.ReadError, Exception)) as ei:
        read.read_manifest(m_bad, dev)
    assert "corrupt" in str(ei.value) or "SHA-256" in str(ei.value)


def test_read_path_at_scale(dev):
    """256 MiB: every stored record decodes, the assembled bytes equal the input, all digests re-verify."""
    import torch
    from hmse_amd import IngestConfig, corpus, ingest, read
    data = torch.from_numpy(corpus.wiki_synth(256 << 20, seed=42)).to(dev)
    res = ingest.ingest_shard(data, IngestConfig())
    back = read.reconstruct_shard(res, verify=True)
    assert torch.equal(back, data)


def test_read_path_refuses_inconsistent_inputs(orc, dev):
    """Out-of-range stream offsets, a DELTA record without a valid base, and a chunk map that disagrees with the stored
    lengths are reported (per-record flag / HmseError), never dereferenced."""
    import torch, zlib
    from hmse_amd import ops
    c = zlib.compressobj(9, zlib.DEFLATED, -15)
    s = c.compress(b"hello hello hello hello") + c.flush()
    streams = torch.from_numpy(np.frombuffer(s * 3, np.uint8).copy()).to(dev)
    n = len(s)
    off = torch.tensor([0, n, 2 * n, 3 * n], dtype=torch.int64, device=dev)
    kind = torch.tensor([0, 2, 2], dtype=torch.uint8, device=dev)
    raw_len = torch.tensor([23, 23, 23], dtype=torch.int64, device=dev)
    # record 1: DELTA whose base is itself (not earlier); record 2: DELTA with base -1
    base = torch.tensor([-1, 1, -1], dtype=torch.int64, device=dev)
    raw, raw_off, ok = ops.l1_inflate(streams, off, kind, base, raw_len, check=False)
    assert ok.tolist() == [1, 0, 0] and bytes(raw[:23].tolist()) == b"hello hello hello hello"
    with pytest.raises(ops.HmseError):
        ops.l1_inflate(streams, off, kind, base, raw_len)
    # stream positions beyond the blob (manifest-style starts + lengths)
    starts = torch.tensor([0, 10 * n, 2 * n], dtype=torch.int64, device=dev)
    lens = torch.tensor([n, n, 4 * n], dtype=torch.int32, device=dev)
    _, _, ok = ops.l1_inflate(streams, starts, torch.zeros(3, dtype=torch.uint8, device=dev), None, raw_len, stream_len=lens, check=False)
    assert ok.tolist() == [1, 0, 0]
    # chunk map vs stored lengths
    cuts = torch.tensor([0, 23, 46], dtype=torch.int64, device=dev)
    good = ops.read_assemble(cuts, torch.tensor([0, 0], dtype=torch.int64, device=dev), raw_off[:2].contiguous(), raw[:23].contiguous())
    assert bytes(good.tolist()) == b"hello hello hello hello" * 2
    with pytest.raises(ops.HmseError):
        ops.read_assemble(cuts, torch.tensor([0, 5], dtype=torch.int64, device=dev), raw_off[:2].contiguous(), raw[:23].contiguous())
    with pytest.raises(ops.HmseError):
        ops.read_assemble(torch.tensor([0, 20, 46], dtype=torch.int64, device=dev), torch.tensor([0, 0], dtype=torch.int64, device=dev),
                          raw_off[:2].contiguous(), raw[:23].contiguous())


def test_inflate_many_small_records_auto_dispatch(orc, dev):
    """60 000 records (more than the 49 152 from which hmse_l1_inflate picks the lane-per-stream decoder by itself): sizes
    0..3000, zlib levels 0/1/6/9 and strategies mixed, 15 % with an earlier record as dictionary (chains included), 3 %
    corrupted — bytes and accept/reject decisions equal the oracle's under the automatic choice and under both forced decoders
    (this module's fixture forces one; the automatic run is made explicitly)."""
    import zlib
    from hmse_amd import ops
    rnd = random.Random(11)
    text = words_text(1 << 20, seed=5).tobytes()
    recs, raws = [], []
    for k in range(60000):
        n = rnd.choice((0, 1, 5, 40, 300, 1200, 3000)) if rnd.random() < 0.3 else rnd.randrange(3000)
        o = rnd.randrange(len(text) - 3000)
        t = text[o:o + n]
        b = -1
        if k > 10 and rnd.random() < 0.15:
            b = rnd.randrange(max(0, k - 200), k)
            while recs[b][1] == 0 and b > 0:
                b -= 1
        lvl, strat = rnd.choice((0, 1, 6, 9)), rnd.choice((zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE))
        zd = raws[b] if b >= 0 and recs[b][1] else None
        if zd is None:
            b = -1
        c = zlib.compressobj(lvl, zlib.DEFLATED, -15, 9, strat, zdict=zd) if zd else zlib.compressobj(lvl, zlib.DEFLATED, -15, 9, strat)
        s = c.compress(t) + c.flush()
        if rnd.random() < 0.03:
            s = mutate(s, rnd)
        recs.append((s, len(t), b))
        raws.append(t)
    ok = run_both(orc, dev, recs)
    assert 0.9 * len(recs) < ok.sum() < len(recs)
    ops.l1_inflate_mode(0)
    ok0 = run_both(orc, dev, recs)
    assert np.array_equal(ok, ok0)
