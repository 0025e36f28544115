"""CPU: host-side logic — corpus generator, manifest records (the reference's packed structs), stats."""
import hashlib
import os
import zlib

import numpy as np
import pytest
import torch


def test_corpus_blocks_are_independent_and_deterministic():
    from hmse_amd import corpus
    a = corpus.wiki_synth(4 << 20, seed=42)
    b = corpus.wiki_synth(2 << 20, seed=42, first_block=2)           # any rank can generate any block range
    assert np.array_equal(a[2 << 20:], b)
    assert np.array_equal(corpus.wiki_synth(1 << 20, seed=42, threads=1), a[: 1 << 20])
    assert not np.array_equal(corpus.wiki_synth(1 << 20, seed=43), a[: 1 << 20])
    assert a.min() >= 10 and a.max() < 128                             # ASCII text
    r = corpus.random_bytes(4096)
    assert np.array_equal(r, corpus.random_bytes(4096))                # seed 0xDEADBEEF (VALIDATION_METHODS.md:213)
    for prof in ("arxiv", "news", "code"):
        assert corpus.wiki_synth(1 << 20, profile=prof).size == 1 << 20


def test_corpus_has_the_redundancy_profile(orc):
    """Exact duplicates and near-duplicates exist at chunk granularity (README.md:2123-2127)."""
    from hmse_amd import corpus
    d = corpus.wiki_synth(48 << 20, seed=42)
    cfg = orc.default_cfg()
    cuts = orc.cdc(d, cfg)
    fo, _ = orc.dedup(orc.sha256_chunks(d, cuts))
    sz = np.diff(cuts.astype(np.int64))
    dup_bytes = sz[fo != np.arange(len(fo))].sum() / sz.sum()
    assert 0.01 < dup_bytes < 0.4
    z = sum(len(zlib.compress(d[int(cuts[i]):int(cuts[i + 1])].tobytes(), 9)) for i in range(200))
    assert sz[:200].sum() / z > 2.0                                     # README.md:2425 minimum L1 ratio


def _shard_result_from_oracle(orc, data, cfg_kw=None):
    from dataclasses import asdict
    from hmse_amd import IngestConfig, ingest
    cfg = IngestConfig(**(cfg_kw or {}))
    oc = orc.default_cfg(**asdict(cfg))
    cuts = orc.cdc(data, oc)
    dg = orc.sha256_chunks(data, cuts)
    fo, rc = orc.dedup(dg)
    uniq = np.nonzero(fo == np.arange(len(fo)))[0].astype(np.uint64)
    sig = orc.minhash_chunks(data, cuts, oc, uniq)
    keys, base = orc.lsh(sig, oc)
    out, off, kind = orc.deflate_chunks(data, cuts, oc, uniq, base)
    t = torch.from_numpy
    res = ingest.ShardResult(data.size, t(cuts.astype(np.int64)), t(dg), 0, len(cuts) - 1, t(fo.astype(np.int64)), t(rc.astype(np.int32)),
                             t(uniq.astype(np.int64)), t(sig.view(np.int32)), t(keys.view(np.int32)), t(base), t(out), t(off.astype(np.int64)), t(kind))
    res.stats = ingest.shard_stats(res)
    return res


def test_manifest_records_and_reconstruct(orc):
    """ChunkIndex 40 B / DeltaChunk 8 B header / pointer 8 B (README.md:1263-1270, 2182-2189, 1312) and the
    three-branch read path reconstructing the input bit for bit (VALIDATION_METHODS.md:257)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import variants_dataset
    from hmse_amd import corpus, manifest
    from hmse_amd.config import KIND_DELTA, KIND_POINTER
    data = variants_dataset(corpus.wiki_synth(3 << 20, seed=42))
    res = _shard_result_from_oracle(orc, data, {"seg_size": 1 << 20})
    import manifest_ref
    m = manifest_ref.build(res)
    assert m.index.dtype.itemsize == 40 and m.pointers.dtype.itemsize == 8
    assert len(m.index) == res.stats["unique"] and len(m.pointers) == res.stats["pointer"] > 0
    assert (m.chunk_map["kind"] == KIND_DELTA).sum() == res.stats["delta"] > 0
    assert (m.chunk_map["kind"] == KIND_POINTER).sum() == res.stats["pointer"]
    e = m.index[0]
    s0 = m.blob[int(e["lba"]) * m.lba_unit: int(e["lba"]) * m.lba_unit + int(e["length"])].tobytes()
    assert hashlib.sha256(zlib.decompress(s0, -15)).digest() == e["sha256"].tobytes()
    assert int(m.index["refcount"].sum()) == res.stats["chunks"]
    blob = m.to_bytes()
    m2 = manifest.Manifest.from_bytes(blob)
    assert manifest.reconstruct(m2) == data.tobytes()
    assert hashlib.sha256(manifest.reconstruct(m)).digest() == hashlib.sha256(data.tobytes()).digest()


def test_stats_and_cf_formula(orc):
    from hmse_amd import corpus, ingest
    data = corpus.wiki_synth(2 << 20, seed=42)
    res = _shard_result_from_oracle(orc, data)
    st = ingest.merge_stats([res.stats, res.stats])
    s = res.stats
    assert st["bytes"] == 2 * data.size
    # CF = N_in / (stored + 40*unique + 8*pointer + 8*delta)   (VALIDATION_METHODS.md:255)
    assert st["cf"] == pytest.approx(data.size / (s["stored_bytes"] + 40 * s["unique"] + 8 * s["pointer"] + 8 * s["delta"]))
    assert st["cf_payload"] > st["cf"] > 1.5


def test_fixed_cuts_for_l1_only_ablation():
    from hmse_amd import IngestConfig, ingest
    cfg = IngestConfig()
    so = torch.tensor([0, 100_000, 100_000, 170_000])
    c = ingest.fixed_cuts(170_000, cfg, so).tolist()
    assert c[0] == 0 and c[-1] == 170_000 and 100_000 in c and max(np.diff(c)) <= cfg.max_size


def test_band_tables_on_disk_round_trip(orc):
    """README.md:1937-1945 bucket lists: every id sits in the bucket its key names, in ingest order; the LSH base of a
    chunk is the earliest member of one of its buckets; crowded buckets continue across headers."""
    from hmse_amd import bandtable
    rng = np.random.default_rng(3)
    cfg = orc.default_cfg()
    n = 3000
    sig = rng.integers(0, 2**32, (n, 128), dtype=np.uint64).astype(np.uint32)
    sig[500:900, :32] = sig[17, :32]            # a crowded bucket in band 0
    sig[1200:1260, 96:] = sig[40, 96:]          # and one in band 3
    keys, base = orc.lsh(sig, cfg)
    buf = bandtable.write_band_tables(keys, 16)
    bits, tables = bandtable.read_band_tables(buf)
    assert bits == 16 and len(tables) == 4
    assert len(buf) == 8 + 24 + sum(8 + 4 * len(t[0]) + 3 * n for t in tables)   # 4 B per bucket + 3 B per id per band
    for b, (h, start, cnt, ids) in enumerate(tables):
        assert cnt.sum() == n and (np.diff(h.astype(np.int64)) > 0).all()
        for j in (0, len(h) // 2, len(h) - 1):
            members = ids[start[j]: start[j] + cnt[j]]
            assert ((keys[members, b] & 0xFFFF) == h[j]).all() and (np.diff(members.astype(np.int64)) > 0).all()
    for i in np.nonzero(base >= 0)[0][:200]:
        firsts = [bandtable.candidates(tables, b, int(keys[i, b] & 0xFFFF)) for b in range(4)]
        assert any(base[i] in f for f in firsts) and base[i] < i
    assert base[600] == 17 and bandtable.candidates(tables, 0, int(keys[17, 0] & 0xFFFF))[0] <= 17
    # > 65535 ids in one bucket: continuation headers
    big = np.zeros((70000, 4), np.uint32); big[:, 1] = np.arange(70000)
    _, t2 = bandtable.read_band_tables(bandtable.write_band_tables(big, 16))
    assert t2[0][2].tolist() == [70000] and np.array_equal(bandtable.candidates(t2, 0, 0), np.arange(70000))


def test_parse_manifest_matches_the_oracle_records(orc):
    """Host half of the GPU read path (hmse_amd/read.py): kinds, dictionary slots, stream positions and raw lengths decoded
    from the packed records equal what the writer was given; the oracle inflate decodes them to the input."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import variants_dataset
    from hmse_amd import corpus, manifest, read
    data = variants_dataset(corpus.wiki_synth(3 << 20, seed=42))
    res = _shard_result_from_oracle(orc, data, {"seg_size": 1 << 20})
    import manifest_ref
    m = manifest.Manifest.from_bytes(manifest_ref.build(res).to_bytes())
    p = read.parse_manifest(m)
    kind, base = res.kind.numpy(), res.base.numpy()
    off = res.stream_off.numpy()
    assert np.array_equal(p["kind"], kind)
    assert np.array_equal(p["base"], np.where(kind == 2, base, -1))
    assert np.array_equal(p["stream_len"], np.diff(off))
    lens = np.diff(res.cuts.numpy())
    assert np.array_equal(p["raw_len"], lens[res.uniq_ids.numpy()])
    raw, raw_off, ok = orc.inflate_chunks(m.blob, p["stream_off"], p["kind"], p["base"], p["raw_len"], stream_len=p["stream_len"])
    assert ok.all()
    want = np.concatenate([data[int(res.cuts[i]):int(res.cuts[i + 1])] for i in res.uniq_ids.tolist()])
    assert np.array_equal(raw, want)
    # a header that names an LBA outside the index is refused
    bad = m.blob.copy()
    k = int(np.nonzero(kind == 2)[0][0]); o = int(m.index["lba"][k]) * m.lba_unit
    bad[o:o + 4] = np.frombuffer(np.uint32(0xFFFFFFF0).tobytes(), np.uint8)
    with pytest.raises(read.ReadError):
        read.parse_manifest(manifest.Manifest(m.lba_unit, m.index, m.chunk_map, m.pointers, bad))


def test_oracle_pool_equals_the_serial_oracle_pipeline(orc):
    """tests/oracle_pool.py (the oracle with MinHash and DEFLATE spread over spawned workers, used by the at-scale GPU
    parity test) returns exactly what the serial pipeline returns, slices and cross-slice dictionaries included."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import oracle_pool
    from make_golden import variants_dataset
    from test_gpu_ingest import oracle_pipeline
    from hmse_amd import IngestConfig, corpus
    cfg = IngestConfig(seg_size=1 << 20)
    data = np.concatenate([variants_dataset(corpus.wiki_synth(3 << 20, seed=42)), corpus.wiki_synth(1 << 20, seed=42)])
    a = oracle_pool.pipeline(orc, data, cfg, workers=2, slices_per_worker=6)
    _, (b,) = oracle_pipeline(orc, data, cfg)
    assert (b["base"] >= 0).sum() > 10
    for k in ("cuts", "dg", "fo", "uniq", "sig", "base", "out", "off", "kind"):
        assert np.array_equal(a[k], b[k]), k


def test_two_shard_store_merges_and_reconstructs(orc):
    """A sharded store: one manifest per shard, duplicates whose first occurrence lives on the other shard
    become POINTERs that name (shard, lba) once merged (README.md:1312, 1635-1669); the stock-zlib verifier reads the whole
    corpus back through them."""
    import manifest_ref
    from test_gpu_ingest import oracle_pipeline
    from hmse_amd import IngestConfig, corpus, ingest, manifest
    from hmse_amd.config import KIND_POINTER
    cfg = IngestConfig(seg_size=1 << 20)
    data = corpus.wiki_synth(6 << 20, seed=42)
    data[(4 << 20) + 5000:(5 << 20)] = data[5000:(1 << 20)]                       # cross-shard duplicates
    shards, want = oracle_pipeline(orc, data, cfg, n_shards=2)
    t = torch.from_numpy
    parts = []
    bases = [w["chunk_base"] for w in want]
    for r, w in enumerate(want):
        n = len(w["cuts"]) - 1
        res = ingest.ShardResult(shards[r].size, t(w["cuts"].astype(np.int64)), t(w["dg"]), w["chunk_base"], sum(len(x["cuts"]) - 1 for x in want),
                                 t(w["fo"].astype(np.int64)), t(np.ones(n, np.int32)), t(w["uniq"].astype(np.int64)), None, None, t(w["base"]),
                                 t(w["out"]), t(w["off"].astype(np.int64)), t(w["kind"]), shard_bases=bases)
        parts.append(manifest_ref.build(res, r, 2))
    cross = (parts[1].chunk_map["shard"] == 0).sum()
    assert cross > 50 and ((parts[1].pointers["flags"] & manifest.PTR_UNRESOLVED) != 0).sum() == cross
    with pytest.raises(Exception):
        manifest.reconstruct(manifest.Store(parts))                               # unresolved pointers cannot be read
    store = manifest.Store.from_bytes(manifest.merge_manifests(parts).to_bytes())
    p1 = store.shards[1]
    assert ((p1.pointers["flags"] & manifest.PTR_UNRESOLVED) == 0).all()
    x = np.nonzero((p1.chunk_map["kind"] == KIND_POINTER) & (p1.chunk_map["shard"] == 0))[0]
    ptr_of_chunk = np.cumsum(p1.chunk_map["kind"] == KIND_POINTER) - 1
    rec = p1.pointers[ptr_of_chunk[x]]
    assert np.array_equal(rec["target_lba"], store.shards[0].index["lba"][p1.chunk_map["slot"][x]]) and ((rec["flags"] >> 4) == 0).all()
    assert manifest.reconstruct(store) == data.tobytes()


def test_band_table_sidecar_carries_keys_and_signatures():
    from hmse_amd import bandtable
    rng = np.random.default_rng(4)
    keys = rng.integers(0, 2**32, (1000, 4), dtype=np.uint32)
    sig = rng.integers(0, 2**32, (1000, 128), dtype=np.uint32)
    buf = bandtable.write_band_tables(keys, 16, signatures=sig)
    k2, s2 = bandtable.read_signatures(buf)
    assert np.array_equal(k2, keys) and np.array_equal(s2, sig)
    bits, tables = bandtable.read_band_tables(buf)                 # the reference-layout part is unchanged by the trailing section
    assert bits == 16 and sum(int(c.sum()) for _, _, c, _ in tables) == 4000
    assert bandtable.read_signatures(bandtable.write_band_tables(keys, 16)) == (None, None)


def test_energy_model_restatement_reproduces_the_documented_figures():
    """the reference's documentation (tools/README.md:78-86) says 833.3 Wh without compression, 726.4 Wh saved,
    ROI 40.4x for 75 GB / CF 9.375 / 1 Mbps / 5 W; its CLI prints 851.3 / 744.4 / 41.4x because it charges the compression
    energy to the no-compression scenario too.  The corrected restatement gives the documented numbers and a break-even CF
    consistent with them."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import energy_model as em
    r = em.compare(75, 9.375, 1, 5, 0.5, 36)
    assert round(r["plain"]["total_wh"], 1) == 833.3 and round(r["saved_wh"], 1) == 726.4 and round(r["roi"], 1) == 40.4
    assert r["plain"]["compression_wh"] == 0
    be = r["breakeven_cf"]
    at = em.scenario(75, be, 1, 5, 0.5, 36)["total_wh"]
    assert abs(at - r["plain"]["total_wh"]) < 1e-9 and 1.0 < be < 1.03
    assert em.breakeven_cf(0.001, 1000, 5, 0.5, 36) == float("inf")


def test_store_with_dictionaries_in_other_shards(orc):
    """Global L4 (SURVEY.md §8e last sentence) on disk, CPU side: the one-shard oracle run split into three shards' records
    — first occurrences, bases and streams are those of the one-shard run, so some DELTA records name a dictionary stored in
    an EARLIER shard.  Their DeltaChunk headers are written unresolved and listed in the manifest's remote_bases table
    (manifest version 3), the merge fills them from the owning shard's index, and the stock-zlib verifier follows them."""
    import manifest_ref
    from test_gpu_ingest import oracle_pipeline
    from hmse_amd import IngestConfig, corpus, ingest, manifest
    from hmse_amd.config import KIND_DELTA
    cfg = IngestConfig(seg_size=1 << 20)
    data = corpus.wiki_synth(6 << 20, seed=42)
    _, (one,) = oracle_pipeline(orc, data, cfg)
    cuts, uniq, base, off, kind = one["cuts"].astype(np.int64), one["uniq"].astype(np.int64), one["base"], one["off"].astype(np.int64), one["kind"]
    bounds = [0, 2 << 20, 4 << 20, 6 << 20]
    cb = [int(np.searchsorted(cuts, b)) for b in bounds]                          # chunk index of every shard boundary (segment-aligned)
    ub = [int(np.searchsorted(uniq, c)) for c in cb]
    t = torch.from_numpy
    parts, n_remote = [], 0
    for r in range(3):
        c0, c1, u0, u1 = cb[r], cb[r + 1], ub[r], ub[r + 1]
        bg = base[u0:u1].copy()
        loc = np.where((bg >= u0) & (bg < u1), bg - u0, -1)
        n_remote += int(((bg >= 0) & (loc < 0) & (kind[u0:u1] == KIND_DELTA)).sum())
        res = ingest.ShardResult(bounds[r + 1] - bounds[r], t(cuts[c0:c1 + 1] - bounds[r]), t(one["dg"][c0:c1]), c0, len(cuts) - 1,
                                 t(one["fo"][c0:c1].astype(np.int64)), t(np.ones(c1 - c0, np.int32)), t(uniq[u0:u1] - c0), None, None, t(loc),
                                 t(one["out"][off[u0]:off[u1]]), t(off[u0:u1 + 1] - off[u0]), t(kind[u0:u1]), shard_bases=cb[:3],
                                 base_global=t(bg), u_base=u0, u_bases=ub[:3])
        parts.append(manifest_ref.build(res, r, 3))
    assert n_remote >= 3 and sum(m.n_remote() for m in parts) == n_remote and parts[0].n_remote() == 0
    pr = next(m for m in parts if m.n_remote())
    assert manifest.Manifest.from_bytes(pr.to_bytes()).n_remote() == pr.n_remote()                      # version 3 round trip
    with pytest.raises(Exception):
        manifest.reconstruct(manifest.Store(parts))                               # unresolved pointers / headers
    store = manifest.Store.from_bytes(manifest.merge_manifests(parts).to_bytes())
    m2 = next(m for m in store.shards if m.n_remote())
    rb = m2.remote_bases
    hdr = m2.blob[(m2.index["lba"][rb["slot"]].astype(np.int64) * m2.lba_unit)[:, None] + np.arange(4)[None, :]].copy().view("<u4")[:, 0]
    assert np.array_equal(hdr, np.array([store.shards[int(s)].index["lba"][int(b)] for s, b in zip(rb["shard"], rb["base_slot"])], np.uint32))
    assert manifest.reconstruct(store) == data.tobytes()


def test_document_aligned_segments_and_shards():
    """hmse_amd.partition (north_star "shards naturally by document"): segments are runs of whole documents of at most seg_size
    bytes, a longer document is cut every seg_size bytes, shards are runs of whole segments with balanced bytes."""
    import torch
    from hmse_amd import corpus, partition
    data = corpus.wiki_synth(3 << 20, seed=42)
    t = torch.from_numpy(data)
    ds = partition.document_starts(t, slab=700001)          # slab boundaries inside the scan
    assert ds[0] == 0 and (np.diff(ds) > 0).all() and len(ds) > 30
    for p in ds[1:]:
        assert bytes(data[p - 1: p + 2]) == b"\n= "
    assert np.array_equal(ds, partition.document_starts(t))
    seg = 256 << 10
    so = partition.document_seg_off(ds, data.size, seg)
    assert so[0] == 0 and so[-1] == data.size and (np.diff(so) > 0).all() and np.diff(so).max() <= seg
    inner = so[1:-1]
    assert np.isin(inner, ds).all()                         # every boundary is a document start (no document here is longer than seg)
    assert np.diff(so)[:-1].min() > seg // 2                # and segments are well filled
    # a document longer than seg_size is capped
    so2 = partition.document_seg_off(np.array([0, 100, 5000000]), 6000000, 1 << 20)
    assert so2.tolist() == [0, 100, 100 + (1 << 20), 100 + (2 << 20), 100 + (3 << 20), 100 + (4 << 20), 5000000, 6000000]
    for world in (1, 2, 3, 8):
        sh = partition.deal_segments(so, world)
        assert sh[0][0] == 0 and sh[-1][1] == len(so) - 1 and all(a[1] == b[0] for a, b in zip(sh[:-1], sh[1:]))
        sizes = [int(so[b] - so[a]) for a, b in sh]
        assert sum(sizes) == data.size and max(sizes) - min(sizes) <= 2 * seg
    lo, hi, rel = partition.shard_seg_off(so, *partition.deal_segments(so, 2)[1])
    assert rel[0] == 0 and int(rel[-1]) == hi - lo and hi == data.size


def test_dependency_order_of_a_store_whose_dictionaries_sit_on_later_shards():
    """read.dependency_order: records keep their order when every dictionary precedes its dependants; otherwise they are sorted by
    dictionary depth (stable) — the store of a multi-rank stream ingested with global L4 (a dictionary on a later-numbered shard)."""
    from hmse_amd import read
    assert read.dependency_order(np.array([-1, 0, 1, -1, 3], np.int64)) is None
    assert read.dependency_order(np.zeros(0, np.int64)) is None
    base = np.array([4, -1, 0, -1, 3, 2, 1], np.int64)          # 0 <- 4 <- 3 ; 2 <- 0 ; 5 <- 2 ; 6 <- 1
    order, new_of_old = read.dependency_order(base)
    assert sorted(order.tolist()) == list(range(7)) and np.array_equal(new_of_old[order], np.arange(7))
    nb = np.where(base >= 0, new_of_old[np.maximum(base, 0)], -1)[order]
    assert (nb < np.arange(7)).all()                              # every dictionary now precedes its dependants
    depth = {1: 0, 3: 0, 4: 1, 6: 1, 0: 2, 2: 3, 5: 4}
    assert [depth[int(o)] for o in order] == sorted(depth.values())   # by depth, original order inside a depth
    assert order.tolist() == [1, 3, 4, 6, 0, 2, 5]
    with pytest.raises(read.ReadError):
        read.dependency_order(np.array([1, 0], np.int64))         # a cycle


def test_bench_started_bare_with_gpus_2_runs_two_ranks_or_fails():
    """`python bench.py --gpus N` WITHOUT torch.distributed.run must never print an n_gpus-1 line for N GPUs:
    it starts the N ranks itself (child launcher, before torch is imported) or exits non-zero.  HMSE_BENCH_LAUNCH_ONLY=1 stops
    every rank after the rendezvous (gloo, no GPU), rank 0 printing who came."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HMSE_BENCH_LAUNCH_ONLY"] = "1"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, timeout=300)
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stderr.decode()[-2000:])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == [0, 1]
    # a launcher environment that disagrees with --gpus is refused, whatever the order of magnitude
    for ws in ("1", "4"):
        q = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE=ws, RANK="0", LOCAL_RANK="0"),
                           capture_output=True, timeout=120)
        assert q.returncode != 0 and not [ln for ln in q.stdout.decode().splitlines() if ln.startswith("{")]


def test_isa_audit_finds_nothing_unpinned():
    """a static guard for the hazard that hung the GPU in rounds 1 and 3 (a cross-lane read inside a loop the
    compiler structurised as divergent).  tools/isa_audit.py disassembles the shipped library (CPU only); every finding must be
    pinned, with its justification, in tests/golden/isa_audit.json — a new one fails here instead of hanging a GPU box.  Also: the
    library imports no asynchronous memset / memcpy (a captured chain must hold kernel nodes only)."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "hmse_amd", "csrc", "libhmse_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__ as g
        g.build()
    sys.path.insert(0, os.path.join(root, "tools"))
    import isa_audit
    pin = json.load(open(os.path.join(root, "tests", "golden", "isa_audit.json")))
    rep = isa_audit.audit(lib)
    assert len(rep) >= 40 and sum(v["loops"] for v in rep.values()) > 300          # the audit saw the kernels and their loops
    # the regression inputs: the two shapes that hung are recognised when fed back as instruction lists
    hang = [(0, "s_mov_b64", "s[0:1], 0", None), (4, "ds_bpermute_b32", "v1, v2, v3", None), (8, "s_andn2_b64", "exec, exec, s[0:1]", None),
            (12, "s_cbranch_execnz", "65533", 4), (16, "s_endpgm", "", None)]
    f, n_loops, n_div = isa_audit.audit_function(hang)
    assert n_loops == 1 and n_div == 1 and [x[1] for x in f] == ["ds_bpermute_b32"]
    scalar = [(0, "s_mov_b32", "s0, 0", None), (4, "ds_bpermute_b32", "v1, v2, v3", None), (8, "s_cmp_lt_u32", "s0, s1", None),
              (12, "s_cbranch_scc1", "65533", 4), (16, "s_endpgm", "", None)]
    assert isa_audit.audit_function(scalar)[0] == []                                   # the same read in a scalar loop is fine
    bad = []
    for kern, ops in isa_audit.summary(rep).items():
        fam = [k for k in pin["kernels"] if kern == k or (k.endswith("<") and kern.startswith(k))]
        if not fam:
            bad.append((kern, ops, "kernel not pinned"))
            continue
        allowed = pin["kernels"][fam[0]]["allowed"]
        for op, cnt in ops.items():
            if cnt > allowed.get(op, 0):
                bad.append((kern, op, cnt, "pinned: %d" % allowed.get(op, 0)))
    assert not bad, bad
    imports = isa_audit.imported_hip_calls(lib)
    assert not [s for s in imports if "Async" in s or "Graph" in s], imports
    assert set(imports) <= set(pin["hip_imports_allowed"]), sorted(set(imports) - set(pin["hip_imports_allowed"]))


def test_python_side_class_caps_are_the_kernel_source_s():
    """bench.py and the tools price the match kernels' size classes with ops.DEFLATE_CLASS_CAPS; the kernel's own caps are the
    HMSE_TCAP_* defaults of l1_deflate.hip (the device classifies): the two must not drift apart."""
    import os
    import re
    from hmse_amd import ops
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hmse_amd", "csrc", "l1_deflate.hip")).read()
    caps = tuple(int(re.search(r"#define %s (\d+)" % name, src).group(1)) for name in ("HMSE_TCAP_S", "HMSE_TCAP_S2", "HMSE_TCAP_SG", "HMSE_TCAP_SG2"))
    sg3 = int(re.search(r"TCAP_SG3 = (\d+)", src).group(1))
    assert caps + (sg3,) == tuple(ops.DEFLATE_CLASS_CAPS)
