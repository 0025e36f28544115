"""GPU: the streaming front end — batches with overlapped host->HBM copies and a persistent index must
give exactly what one ingest_shard call gives for the concatenated input, and read back to the input."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _dataset():
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import variants_dataset
    from hmse_amd import corpus
    a = corpus.wiki_synth(4 << 20, seed=42)
    # later batches repeat and vary the first one: cross-batch exact duplicates (POINTER) and near-duplicates (DELTA)
    v = variants_dataset(a)
    return np.concatenate([a, v, corpus.wiki_synth((12 << 20) - a.size - v.size, seed=7), a[: (1 << 20) + 12345]])


@pytest.mark.parametrize("pinned", [True, False])
def test_stream_equals_single_shot(dev, pinned):
    import torch
    from hmse_amd import IngestConfig, ingest, read, stream
    cfg = IngestConfig(seg_size=1 << 20)
    data = _dataset()
    whole = ingest.ingest_shard(torch.from_numpy(data).to(dev), cfg)
    st = stream.StreamIngest(cfg, data.size, dev)
    assert data.size > (11 << 20)
    bounds = [0, 4 << 20, 6 << 20, 11 << 20, data.size]          # uneven batches, the last one a partial segment
    for a, b in zip(bounds[:-1], bounds[1:]):
        h = torch.from_numpy(data[a:b].copy())
        st.push(h.pin_memory() if pinned else h)
    res = st.finish()
    for name in ("cuts", "digests", "first_occ", "refcount", "uniq_ids", "sig", "band_keys", "base", "kind", "stream_off", "streams"):
        assert torch.equal(getattr(res, name), getattr(whole, name)), name
    assert res.stats == whole.stats
    # dictionaries and pointers cross batch boundaries
    b1 = int((res.cuts <= (4 << 20)).sum()) - 1                   # chunks of the first batch
    slot_chunk = res.uniq_ids
    crossing = (res.base >= 0) & (slot_chunk >= b1) & (slot_chunk[res.base.clamp(min=0)] < b1) & (res.kind == 2)
    assert int(crossing.sum()) > 10
    later = torch.arange(res.first_occ.numel(), device=dev) >= b1
    assert int(((res.first_occ < b1) & later).sum()) > 10
    # and the read path returns the input
    assert torch.equal(read.reconstruct_shard(res, verify=True), torch.from_numpy(data).to(dev))


def test_stream_rejects_misaligned_batches(dev):
    import torch
    from hmse_amd import IngestConfig, stream
    st = stream.StreamIngest(IngestConfig(seg_size=1 << 20), 4 << 20, dev)
    st.push(torch.zeros((1 << 20) + 5, dtype=torch.uint8))
    with pytest.raises(ValueError):
        st.push(torch.zeros(1 << 20, dtype=torch.uint8))
    with pytest.raises(ValueError):
        stream.StreamIngest(IngestConfig(seg_size=1 << 20), 1 << 20, dev).push(torch.zeros(2 << 20, dtype=torch.uint8))


def test_resume_from_manifest_then_append(dev):
    """Store A, keep only its manifest bytes, resume from them (GPU read-back restores the history), append B: every output
    equals one ingest of A+B — incremental ingest against an existing index."""
    import torch
    from hmse_amd import IngestConfig, ingest, manifest, read, stream
    cfg = IngestConfig(seg_size=1 << 20)
    data = _dataset()
    split = 6 << 20
    first = ingest.ingest_shard(torch.from_numpy(data[:split]).to(dev), cfg)
    blob = manifest.build_manifest(first).to_bytes()
    del first
    st = stream.StreamIngest.resume(manifest.Manifest.from_bytes(blob), cfg, data.size, dev)
    st.push(torch.from_numpy(data[split:].copy()))
    res = st.finish()
    whole = ingest.ingest_shard(torch.from_numpy(data).to(dev), cfg)
    for name in ("cuts", "digests", "first_occ", "refcount", "uniq_ids", "sig", "band_keys", "base", "kind", "stream_off", "streams"):
        assert torch.equal(getattr(res, name), getattr(whole, name)), name
    assert torch.equal(read.reconstruct_shard(res, verify=True), torch.from_numpy(data).to(dev))


def test_resume_loads_the_index_from_chunkindex_and_band_table_sidecar(dev):
    """resume() with the band-table sidecar: digests go from the ChunkIndex records into the L3 table, band keys and
    signatures from the sidecar into the L4 tables — nothing is re-hashed (no SHA-256 verify pass, no MinHash) — and the
    appended batches still equal one ingest of the whole stream, bit for bit (README.md:1937-1945)."""
    import torch
    from hmse_amd import IngestConfig, bandtable, ingest, manifest, ops, read, stream
    cfg = IngestConfig(seg_size=1 << 20)
    data = _dataset()
    split = 6 << 20
    st0 = stream.StreamIngest(cfg, split, dev)
    st0.push(torch.from_numpy(data[:split].copy()))
    first = st0.finish()
    blob = manifest.build_manifest(first).to_bytes()
    side = st0.index_sidecar()
    n_first = first.uniq_ids.numel()
    del st0, first
    keys, sig = bandtable.read_signatures(side)
    assert sig.shape == (n_first, 128) and keys.shape == (n_first, 4)
    calls = {"minhash": 0, "sha": 0}
    real_mh, real_sha = ops.l4_minhash, ops.l3_sha256
    def count(name, fn):
        def w(*a, **k):
            calls[name] += 1
            return fn(*a, **k)
        return w
    ops.l4_minhash, ops.l3_sha256 = count("minhash", real_mh), count("sha", real_sha)
    try:
        st = stream.StreamIngest.resume(manifest.Manifest.from_bytes(blob), cfg, data.size, dev, band_tables=side, verify=False)
        assert calls == {"minhash": 0, "sha": 0}                      # the history was loaded, not re-hashed
    finally:
        ops.l4_minhash, ops.l3_sha256 = real_mh, real_sha
    st.push(torch.from_numpy(data[split:split + (2 << 20)].copy()))
    st.push(torch.from_numpy(data[split + (2 << 20):].copy()))
    res = st.finish()
    whole = ingest.ingest_shard(torch.from_numpy(data).to(dev), cfg)
    for name in ("cuts", "digests", "first_occ", "refcount", "uniq_ids", "sig", "band_keys", "base", "kind", "stream_off", "streams"):
        assert torch.equal(getattr(res, name), getattr(whole, name)), name
    assert torch.equal(read.reconstruct_shard(res, verify=True), torch.from_numpy(data).to(dev))
    with pytest.raises(ValueError):
        stream.StreamIngest.resume(manifest.Manifest.from_bytes(blob), cfg, data.size, dev, band_tables=bandtable.write_band_tables(keys[:5], 16, sig[:5]))


def test_captured_chain_equals_eager_and_one_shot(dev):
    """BASELINE configs[4] "hipGraph-captured per-batch pipeline": with graph=True every batch is ONE enqueue of the
    device-count chain (no host read between stages); the first batch of a size is enqueued directly, the second is
    captured into a hipGraph, the third and later ones replay it.  Every output equals the host-sized eager path and one
    ingest of the whole stream, bit for bit — also across a batch of a different size and after resume()."""
    import torch
    from hmse_amd import IngestConfig, ingest, manifest, read, stream
    cfg = IngestConfig(seg_size=1 << 20)
    data = _dataset()
    whole = ingest.ingest_shard(torch.from_numpy(data).to(dev), cfg)
    names = ("cuts", "digests", "first_occ", "refcount", "uniq_ids", "sig", "band_keys", "base", "kind", "stream_off", "streams")
    B = 2 << 20
    for graph in (True, False):
        st = stream.StreamIngest(cfg, data.size, dev, graph=graph)
        for a in range(0, data.size, B):
            st.push(torch.from_numpy(data[a: a + B].copy()))
        res = st.finish()
        for name in names:
            assert torch.equal(getattr(res, name), getattr(whole, name)), (graph, name)
        if graph:
            sizes = {k: (v[0] is not None) for k, v in st._graphs.items()}
            assert sizes[B] is True and len(sizes) == 2          # the common size was captured and replayed; the ragged tail ran directly
            assert torch.equal(read.reconstruct_shard(res, verify=True), torch.from_numpy(data).to(dev))
    # resume() + captured chain
    split = 6 << 20
    first = ingest.ingest_shard(torch.from_numpy(data[:split]).to(dev), cfg)
    m = manifest.Manifest.from_bytes(manifest.build_manifest(first).to_bytes())
    del first
    st = stream.StreamIngest.resume(m, cfg, data.size, dev, graph=True)
    for a in range(split, data.size, 1 << 20):
        st.push(torch.from_numpy(data[a: a + (1 << 20)].copy()))
    res = st.finish()
    for name in names:
        assert torch.equal(getattr(res, name), getattr(whole, name)), ("resume", name)


def _oracle_windowed(orc, data, cfg, window_starts):
    """CPU restatement of a windowed stream: chunking and exact dedupe over the whole stream; MinHash/LSH bases and dictionary DEFLATE
    per WINDOW (a dictionary is a stored chunk of the same window)."""
    from dataclasses import asdict
    oc = orc.default_cfg(**asdict(cfg))
    cuts = orc.cdc(data, oc)
    dg = orc.sha256_chunks(data, cuts)
    fo, rc = orc.dedup(dg)
    uniq = np.nonzero(fo == np.arange(len(fo)))[0].astype(np.uint64)
    sig = orc.minhash_chunks(data, cuts, oc, uniq)
    base = np.full(len(uniq), -1, np.int64)
    kind = np.zeros(len(uniq), np.uint8)
    parts, lens = [], []
    bounds = list(window_starts) + [data.size]
    for a, b in zip(bounds[:-1], bounds[1:]):
        sel = np.nonzero((cuts[uniq] >= a) & (cuts[uniq] < b))[0]
        if not len(sel):
            continue
        _, bw = orc.lsh(sig[sel], oc)
        base[sel] = np.where(bw >= 0, sel[np.maximum(bw, 0)], -1)
        out, off, kd = orc.deflate_chunks(data, cuts, oc, uniq[sel], bw)
        parts.append(out); lens.append(np.diff(off.astype(np.int64))); kind[sel] = kd
    off = np.concatenate([[0], np.cumsum(np.concatenate(lens))]).astype(np.uint64)
    return dict(cuts=cuts, dg=dg, fo=fo, rc=rc, uniq=uniq, sig=sig, base=base, kind=kind, off=off, out=np.concatenate(parts))


def test_windowed_stream_keeps_only_a_window_resident_and_equals_the_oracle_with_the_same_windows(orc, dev):
    """VERDICT r3 missing 5 / item 9: a stream longer than the resident buffer (README.md:262-276, 1579-1580: the reference's loop runs in
    fixed memory "until input exhausted").  StreamIngest(window_bytes=W): the raw bytes of a window replace the previous window's, the band
    tables are cleared at a window's start (a dictionary is a chunk of the same window), exact dedupe still spans the whole stream.  Every
    output equals the CPU oracle with the same windows; the buffer holds W bytes, not the stream; the records decode through stock zlib."""
    import zlib
    import torch
    from hmse_amd import IngestConfig, ingest, stream
    cfg = IngestConfig(seg_size=1 << 20)
    data = _dataset()
    B, W = 2 << 20, 4 << 20
    s = stream.StreamIngest(cfg, data.size, dev, graph=True, window_bytes=W)
    assert s.data.numel() == W                                              # the resident buffer is the window, not the stream
    for a in range(0, data.size, B):
        s.push(torch.from_numpy(data[a: a + B].copy()))
    res = s.finish()
    torch.cuda.synchronize()
    assert s.window_starts == list(range(0, data.size, W))
    o = _oracle_windowed(orc, data, cfg, s.window_starts)
    assert np.array_equal(res.cuts.cpu().numpy().astype(np.uint64), o["cuts"])
    assert np.array_equal(res.digests.cpu().numpy(), o["dg"])
    assert np.array_equal(res.first_occ.cpu().numpy().astype(np.uint64), o["fo"])          # dedupe across windows
    assert np.array_equal(res.uniq_ids.cpu().numpy().astype(np.uint64), o["uniq"])
    assert np.array_equal(res.sig.cpu().numpy().view(np.uint32), o["sig"])
    assert np.array_equal(res.base.cpu().numpy(), o["base"])
    assert np.array_equal(res.kind.cpu().numpy(), o["kind"])
    assert np.array_equal(res.stream_off.cpu().numpy().astype(np.uint64), o["off"])
    assert np.array_equal(res.streams.cpu().numpy(), o["out"])
    # bases never cross a window; the unwindowed stream finds more of them (the variants of batch 0 arrive in later windows)
    cuts, uniq, base = o["cuts"], o["uniq"], o["base"]
    hb = base >= 0
    wof = lambda c: np.searchsorted(np.array(s.window_starts), cuts[c], side="right")
    assert hb.sum() > 20 and (wof(uniq[hb]) == wof(uniq[base[hb]])).all()
    whole = ingest.ingest_shard(torch.from_numpy(data).to(dev), cfg)
    assert int((whole.base >= 0).sum()) > int(hb.sum())
    assert (o["fo"] != np.arange(len(o["fo"]))).sum() > 100 and (cuts[o["fo"][-50:].astype(np.int64)] < W).any()   # a POINTER into a window long gone
    out, off, kind = res.streams.cpu().numpy(), o["off"].astype(np.int64), o["kind"]
    for k in range(0, len(uniq), 5):
        c = int(uniq[k]); zd = None
        if kind[k] == 2:
            b = int(uniq[base[k]]); zd = data[int(cuts[b]):int(cuts[b + 1])].tobytes()
        d = zlib.decompressobj(-15, zdict=zd) if zd else zlib.decompressobj(-15)
        assert d.decompress(out[off[k]:off[k + 1]].tobytes()) == data[int(cuts[c]):int(cuts[c + 1])].tobytes(), k
