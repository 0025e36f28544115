"""GPU: the request-driven read path (README.md:1444-1448 "Read Request (offset, len)" -> chunk map; 1621-1675 three branches;
gate README.md:1329 "100 % checksum pass for 1000 random articles"): read.StoreReader decodes only the closure of the
requested chunks — POINTER targets, DELTA dictionaries, dictionaries of dictionaries — and returns exactly the input's bytes."""
import dataclasses

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _data():
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import variants_dataset
    from hmse_amd import corpus
    a = corpus.wiki_synth(3 << 20, seed=42)
    # a family of successive light edits of one region: every member is a near-duplicate of the first (its LSH base)
    rng = np.random.default_rng(5)
    fam, v = [], a[500000: 500000 + 160000].copy()
    for _ in range(6):
        v = v.copy()
        v[rng.integers(0, v.size, 12)] = rng.integers(97, 123, 12, dtype=np.uint8)
        fam.append(v)
    return np.concatenate([a, variants_dataset(a), np.concatenate(fam), corpus.wiki_synth(2 << 20, seed=9), a[: 700000]])


def _chained(res, data_t, cfg, dev):
    """The same shard with CHAINED dictionaries: wherever the LSH gave b -> a and c -> a (a family of near-duplicates), re-base
    c on b, so that decoding c needs b, which needs a.  Streams re-encoded by the GPU encoder with that base array."""
    import torch
    from hmse_amd import ops
    base = res.base.cpu().numpy().copy()
    order = {}
    n_chain = 0
    for k in range(len(base)):
        if base[k] >= 0:
            prev = order.get(int(base[k]))
            order[int(base[k])] = k
            if prev is not None:
                base[k] = prev          # the previous member of the family instead of its head
                n_chain += 1
    assert n_chain > 20
    b = torch.from_numpy(base).to(dev)
    streams, off, kind = ops.l1_deflate(data_t, res.cuts, cfg, res.uniq_ids, b)
    return dataclasses.replace(res, base=b, streams=streams, stream_off=off, kind=kind), base


def test_1000_random_ranges_equal_the_input_slices(dev):
    import torch
    from hmse_amd import IngestConfig, ingest, manifest, read
    cfg = IngestConfig(seg_size=1 << 20)
    data = _data()
    data_t = torch.from_numpy(data).to(dev)
    res0 = ingest.ingest_shard(data_t, cfg)
    res, base = _chained(res0, data_t, cfg, dev)
    kind = res.kind.cpu().numpy()
    depth = np.zeros(len(base), np.int64)
    for k in range(len(base)):
        if base[k] >= 0 and kind[k] == 2:
            depth[k] = depth[base[k]] + 1
    assert depth.max() >= 3 and res.stats["pointer"] > 50          # chains of dictionaries and POINTER chunks exist
    m = manifest.Manifest.from_bytes(manifest.build_manifest(res).to_bytes())
    rd = read.StoreReader(m, dev)
    assert rd.n_bytes == data.size
    rng = np.random.default_rng(1329)
    off = rng.integers(0, data.size - 1, 1000)
    ln = np.minimum(rng.integers(1, 65536, 1000), data.size - off)         # "articles" of up to 64 KiB
    ln[:5] = [1, 0, 2, data.size - off[3], 1]
    off[4] = data.size - 1
    # one batch of 1000 requests
    got = rd.read_ranges(list(zip(off.tolist(), ln.tolist())))
    for o, n, g in zip(off, ln, got):
        assert g.numel() == n and np.array_equal(g.cpu().numpy(), data[o: o + n]), (o, n)
    assert rd.last["requests"] == 1000 and rd.last["bytes_requested"] == int(ln.sum())
    # request by request: only the closure is decoded — far less than the store — and it contains dictionaries of dictionaries
    pulled = 0
    for o, n in zip(off[:100].tolist(), ln[:100].tolist()):
        (g,) = rd.read_ranges([(o, n)])
        assert np.array_equal(g.cpu().numpy(), data[o: o + n])
        assert rd.last["bytes_decoded"] <= (n + 2 * cfg.max_size) * (depth.max() + 2)
        pulled += rd.last["dictionaries_pulled_in"]
    assert pulled > 10
    # a request inside a chunk whose record is a deep DELTA: the whole chain comes along, nothing else
    k = int(np.argmax(depth))
    c = int(res.uniq_ids[k])
    o = int(res.cuts[c]) + 5
    (g,) = rd.read_ranges([(o, 100)])
    assert np.array_equal(g.cpu().numpy(), data[o: o + 100])
    assert rd.last["records_decoded"] == depth[k] + 1 and rd.last["chunks_touched"] == 1
    with pytest.raises(read.ReadError):
        rd.read_ranges([(data.size - 10, 11)])


def test_ranges_over_a_two_shard_store_cross_shard_pointers(dev):
    import torch
    from hmse_amd import IngestConfig, ingest, manifest, read
    cfg = IngestConfig(seg_size=1 << 20)
    data = _data()[: 6 << 20].copy()
    data[(4 << 20) + 5000: (5 << 20)] = data[5000: (1 << 20)]                  # the second shard repeats bytes of the first
    shards = [torch.from_numpy(data[: 3 << 20]).to(dev), torch.from_numpy(data[3 << 20:]).to(dev)]
    res = ingest.ingest_shards_local(shards, cfg)
    store = manifest.merge_manifests([manifest.build_manifest(r, i, 2) for i, r in enumerate(res)])
    rd = read.StoreReader(manifest.Store.from_bytes(store.to_bytes()), dev)
    rng = np.random.default_rng(7)
    off = rng.integers(0, data.size - 70000, 300)
    ln = rng.integers(1, 70000, 300)
    off[0], ln[0] = (3 << 20) - 1000, 5000                                        # straddles the shard boundary
    off[1], ln[1] = (4 << 20) + 6000, 60000                                       # chunks stored in the other shard
    for o, n, g in zip(off, ln, rd.read_ranges(list(zip(off.tolist(), ln.tolist())))):
        assert np.array_equal(g.cpu().numpy(), data[o: o + n]), (o, n)
