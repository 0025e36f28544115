"""GPU: the whole hot path through the host driver, against the oracle pipeline and through size-independent
properties at larger sizes (round trips, determinism, sortedness, shard-count invariance)."""
import hashlib
import zlib
from dataclasses import asdict

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def oracle_pipeline(orc, data, cfg, n_shards=1):
    """CPU restatement of ingest_shard for n_shards contiguous segment runs: global dedupe, local LSH/DEFLATE."""
    oc = orc.default_cfg(**asdict(cfg))
    seg = cfg.seg_size
    nseg = max(n_shards, data.size // seg)
    bounds = [(r * nseg // n_shards) * seg for r in range(n_shards)] + [data.size]
    shards = [data[bounds[r]:bounds[r + 1]] for r in range(n_shards)]
    cuts = [orc.cdc(s, oc) for s in shards]
    dg = [orc.sha256_chunks(s, c) for s, c in zip(shards, cuts)]
    fo, rc = orc.dedup(np.concatenate(dg))
    res, base0 = [], 0
    for r in range(n_shards):
        n = len(cuts[r]) - 1
        fol = fo[base0:base0 + n]
        uniq = np.nonzero(fol == np.arange(base0, base0 + n))[0].astype(np.uint64)
        sig = orc.minhash_chunks(shards[r], cuts[r], oc, uniq)
        keys, base = orc.lsh(sig, oc)
        out, off, kind = orc.deflate_chunks(shards[r], cuts[r], oc, uniq, base)
        res.append(dict(cuts=cuts[r], dg=dg[r], fo=fol, uniq=uniq, sig=sig, base=base, out=out, off=off, kind=kind, chunk_base=base0))
        base0 += n
    return shards, res


def test_full_pipeline_equals_oracle_and_reconstructs(orc, dev):
    import sys, os, torch
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import variants_dataset
    from hmse_amd import IngestConfig, corpus, ingest, manifest
    cfg = IngestConfig(seg_size=1 << 20)
    data = np.concatenate([variants_dataset(corpus.wiki_synth(3 << 20, seed=42)), corpus.wiki_synth(2 << 20, seed=42)])
    res = ingest.ingest_shard(torch.from_numpy(data).to(dev), cfg)
    _, (o,) = oracle_pipeline(orc, data, cfg)
    assert np.array_equal(res.cuts.cpu().numpy().astype(np.uint64), o["cuts"])
    assert np.array_equal(res.digests.cpu().numpy(), o["dg"])
    assert np.array_equal(res.first_occ.cpu().numpy().astype(np.uint64), o["fo"])
    assert np.array_equal(res.uniq_ids.cpu().numpy().astype(np.uint64), o["uniq"])
    assert np.array_equal(res.sig.cpu().numpy().view(np.uint32), o["sig"])
    assert np.array_equal(res.base.cpu().numpy(), o["base"])
    assert np.array_equal(res.kind.cpu().numpy(), o["kind"])
    assert np.array_equal(res.stream_off.cpu().numpy().astype(np.uint64), o["off"])
    assert np.array_equal(res.streams.cpu().numpy(), o["out"])
    assert res.stats["delta"] > 10 and res.stats["pointer"] > 10
    m = manifest.build_manifest(res)
    assert manifest.reconstruct(manifest.Manifest.from_bytes(m.to_bytes())) == data.tobytes()   # 100 % lossless (VALIDATION_METHODS.md:257)
    # the GPU packer (hmse_manifest_pack) writes exactly the bytes of the host reference writer
    import manifest_ref
    assert manifest_ref.build(_to_host(res)).to_bytes() == m.to_bytes()


@pytest.mark.parametrize("preset", ["l1_only", "l1_cdc", "l1_cdc_dedupe", "full", "l4_only", "cdc_dedupe"])
def test_ablation_matrix_runs_and_is_lossless(preset, dev):
    """VALIDATION_METHODS.md:458-464: every layer subset produces a decodable result."""
    import torch
    from hmse_amd import ABLATIONS, IngestConfig, corpus, ingest
    cfg = IngestConfig(layers=ABLATIONS[preset], seg_size=1 << 20)
    data = corpus.wiki_synth(2 << 20, seed=42)
    res = ingest.ingest_shard(torch.from_numpy(data).to(dev), cfg)
    cuts = res.cuts.cpu().numpy()
    assert cuts[0] == 0 and cuts[-1] == data.size and (np.diff(cuts) > 0).all() and np.diff(cuts).max() <= cfg.max_size
    if res.streams is not None:
        off = res.stream_off.cpu().numpy(); out = res.streams.cpu().numpy(); kind = res.kind.cpu().numpy()
        uniq = res.uniq_ids.cpu().numpy(); base = res.base.cpu().numpy() if res.base is not None else None
        for k in range(0, len(uniq), 7):
            zd = data[cuts[uniq[base[k]]]:cuts[uniq[base[k]] + 1]].tobytes() if kind[k] == 2 else None
            d = zlib.decompressobj(-15, zdict=zd) if zd else zlib.decompressobj(-15)
            assert d.decompress(out[off[k]:off[k + 1]].tobytes()) == data[cuts[uniq[k]]:cuts[uniq[k] + 1]].tobytes()
    else:
        assert (preset == "l4_only" and res.sig is not None) or (preset == "cdc_dedupe" and res.digests is not None and res.sig is None)


def test_two_shard_run_equals_oracle_with_two_shards(orc, dev):
    """Shard-count invariance without a second GPU: run both shards here, feed the concatenated digests to the
    dedupe op exactly as the all-gather would, compare with the oracle's 2-shard pipeline."""
    import torch
    from hmse_amd import IngestConfig, corpus, ops
    cfg = IngestConfig(seg_size=1 << 20)
    data = corpus.wiki_synth(6 << 20, seed=42)
    data[(4 << 20) + 5000:(5 << 20)] = data[5000:(1 << 20)]                       # cross-shard duplicates
    shards, want = oracle_pipeline(orc, data, cfg, n_shards=2)
    d = [torch.from_numpy(s).to(dev) for s in shards]
    cuts = [ops.l2_cdc(x, cfg) for x in d]
    dg = [ops.l3_sha256(x, c) for x, c in zip(d, cuts)]
    fo, rc = ops.l3_dedup(torch.cat(dg))
    base0 = 0
    n_cross = 0
    for r in range(2):
        n = cuts[r].numel() - 1
        fol = fo[base0:base0 + n]
        assert np.array_equal(cuts[r].cpu().numpy().astype(np.uint64), want[r]["cuts"])
        assert np.array_equal(fol.cpu().numpy().astype(np.uint64), want[r]["fo"])
        uniq = (fol == torch.arange(base0, base0 + n, device=dev)).nonzero().flatten()
        assert np.array_equal(uniq.cpu().numpy().astype(np.uint64), want[r]["uniq"])
        sig = ops.l4_minhash(d[r], cuts[r], cfg, uniq)
        _, base = ops.l4_lsh(sig, cfg)
        out, off, kind = ops.l1_deflate(d[r], cuts[r], cfg, uniq, base)
        assert np.array_equal(base.cpu().numpy(), want[r]["base"]) and np.array_equal(out.cpu().numpy(), want[r]["out"])
        n_cross += int((fol < base0).sum())
        base0 += n
    assert n_cross > 50
    # and the 1-shard cuts are the concatenation of the 2-shard cuts
    c1 = ops.l2_cdc(torch.from_numpy(data).to(dev), cfg).cpu().numpy()
    assert np.array_equal(c1, np.concatenate([want[0]["cuts"], want[1]["cuts"][1:] + np.uint64(shards[0].size)]).astype(np.int64))


def test_properties_at_scale_256MiB(dev):
    """Size-independent properties where the oracle would take minutes: determinism, sortedness, bounds,
    digest spot checks, dedupe idempotence, signature/LSH invariants, DEFLATE round trips."""
    import torch
    from hmse_amd import IngestConfig, corpus, ingest
    cfg = IngestConfig()
    host = corpus.wiki_synth(256 << 20, seed=42)
    data = torch.from_numpy(host).to(dev)
    a = ingest.ingest_shard(data, cfg)
    b = ingest.ingest_shard(data, cfg)
    for f in ("cuts", "digests", "first_occ", "refcount", "uniq_ids", "sig", "band_keys", "base", "streams", "stream_off", "kind"):
        assert torch.equal(getattr(a, f), getattr(b, f)), f                       # reruns bitwise identical
    cuts = a.cuts.cpu().numpy()
    sz = np.diff(cuts)
    assert cuts[0] == 0 and cuts[-1] == host.size and (sz > 0).all() and sz.max() <= cfg.max_size
    assert set(range(cfg.seg_size, host.size, cfg.seg_size)) <= set(cuts.tolist())
    short_ends = cuts[1:][sz < cfg.min_size]
    assert all(int(e) % cfg.seg_size == 0 or int(e) == host.size for e in short_ends)
    rng = np.random.default_rng(0)
    dg = a.digests.cpu().numpy()
    for i in rng.integers(0, len(sz), 300):
        assert dg[i].tobytes() == hashlib.sha256(host[cuts[i]:cuts[i + 1]].tobytes()).digest()
    fo = a.first_occ.cpu().numpy(); rc = a.refcount.cpu().numpy()
    assert (fo <= np.arange(len(fo))).all() and (fo[fo] == fo).all()              # idempotent, points backwards
    assert rc.sum() == len(fo) and (rc[fo != np.arange(len(fo))] == 0).all()
    assert (dg[fo] == dg).all()
    base = a.base.cpu().numpy(); sig = a.sig.cpu().numpy()
    hit = np.nonzero(base >= 0)[0]
    assert (base[hit] < hit).all() and len(hit) > 100
    for k in hit[:200]:                                                            # a base shares a whole band
        assert any((sig[k, 32 * bnd:32 * bnd + 32] == sig[base[k], 32 * bnd:32 * bnd + 32]).all() for bnd in range(4))
    off = a.stream_off.cpu().numpy(); out = a.streams.cpu().numpy(); kind = a.kind.cpu().numpy(); uniq = a.uniq_ids.cpu().numpy()
    assert off[0] == 0 and off[-1] == out.size and (np.diff(off) > 0).all()
    for k in np.concatenate([rng.integers(0, len(uniq), 400), np.nonzero(kind == 2)[0][:100]]):
        zd = host[cuts[uniq[base[k]]]:cuts[uniq[base[k]] + 1]].tobytes() if kind[k] == 2 else None
        d = zlib.decompressobj(-15, zdict=zd) if zd else zlib.decompressobj(-15)
        assert d.decompress(out[off[k]:off[k + 1]].tobytes()) == host[cuts[uniq[k]]:cuts[uniq[k] + 1]].tobytes()
        assert d.eof
    st = ingest.merge_stats([a.stats])
    assert st["cf"] > 2.0 and 0 < st["delta_rate"] < 1


def test_incompressible_input_at_scale(dev):
    """PRNG bytes, seed 0xDEADBEEF (VALIDATION_METHODS.md:213): CF ~ 1, every chunk a stored block.  More jobs than
    the persistent encode grid has workgroups, so a workgroup must survive the stored-block path."""
    import torch
    from hmse_amd import IngestConfig, corpus, ingest, ops
    cfg = IngestConfig()
    host = corpus.random_bytes(96 << 20)
    data = torch.from_numpy(host).to(dev)
    res = ingest.ingest_shard(data, cfg)
    L = (res.cuts[1:] - res.cuts[:-1])[res.uniq_ids]
    ln = res.stream_off[1:] - res.stream_off[:-1]
    assert torch.equal(ln, L + 5) and int((res.kind != 0).sum()) == 0
    off = res.stream_off.cpu().numpy(); out = res.streams.cpu().numpy(); cuts = res.cuts.cpu().numpy()
    for k in (0, 1, len(off) // 2, len(off) - 2):
        assert zlib.decompress(out[off[k]:off[k + 1]].tobytes(), -15) == host[cuts[k]:cuts[k + 1]].tobytes()
    st = ingest.merge_stats([res.stats])
    assert 0.98 < st["cf"] < 1.0


def _to_host(res):
    from hmse_amd import ingest
    c = lambda t: None if t is None else t.cpu()
    return ingest.ShardResult(res.n_bytes, c(res.cuts), c(res.digests), res.chunk_base, res.n_global, c(res.first_occ), c(res.refcount), c(res.uniq_ids),
                              c(res.sig), c(res.band_keys), c(res.base), c(res.streams), c(res.stream_off), c(res.kind), shard_bases=res.shard_bases,
                              base_global=c(getattr(res, "base_global", None)), u_base=getattr(res, "u_base", 0), u_bases=getattr(res, "u_bases", None))


def test_two_shard_store_on_one_gpu_merges_and_reads_back(orc, dev):
    """BASELINE configs[3] as a complete product path, without a second GPU: two shards ingested one after the other with the
    digest exchange an all-gather would deliver, one manifest per shard packed on the GPU, merged into a store whose
    cross-shard POINTER records name (shard, lba) (README.md:1312, 1635-1669), read back on the GPU byte for byte; the
    per-shard results equal the oracle's 2-shard pipeline and the packer equals the host reference writer."""
    import torch
    import manifest_ref
    from hmse_amd import IngestConfig, corpus, ingest, manifest, read
    from hmse_amd.config import KIND_POINTER
    cfg = IngestConfig(seg_size=1 << 20)
    data = corpus.wiki_synth(6 << 20, seed=42)
    data[(4 << 20) + 5000:(5 << 20)] = data[5000:(1 << 20)]                       # cross-shard duplicates
    shards, want = oracle_pipeline(orc, data, cfg, n_shards=2)
    results = ingest.ingest_shards_local([torch.from_numpy(s).to(dev) for s in shards], cfg)
    for r, (res, w) in enumerate(zip(results, want)):
        assert res.chunk_base == w["chunk_base"] and res.shard_bases == [x["chunk_base"] for x in want]
        assert np.array_equal(res.first_occ.cpu().numpy().astype(np.uint64), w["fo"])
        assert np.array_equal(res.uniq_ids.cpu().numpy().astype(np.uint64), w["uniq"])
        assert np.array_equal(res.streams.cpu().numpy(), w["out"]) and np.array_equal(res.kind.cpu().numpy(), w["kind"])
    parts = [manifest.build_manifest(res, r, 2) for r, res in enumerate(results)]
    for r, (res, m) in enumerate(zip(results, parts)):
        assert manifest_ref.build(_to_host(res), r, 2).to_bytes() == m.to_bytes()
    assert ((parts[1].pointers["flags"] & manifest.PTR_UNRESOLVED) != 0).sum() > 50
    with pytest.raises(read.ReadError):
        read.read_store(manifest.Store(parts), dev)
    store = manifest.Store.from_bytes(manifest.merge_manifests(parts).to_bytes())
    assert ((store.shards[1].chunk_map["kind"] == KIND_POINTER) & (store.shards[1].chunk_map["shard"] == 0)).sum() > 50
    back = read.read_store(store, dev, verify=True)
    assert torch.equal(back, torch.from_numpy(data).to(dev))
    assert manifest.reconstruct(store) == data.tobytes()                          # and through stock zlib on the host
    # a single-shard store reads back through the same entry point
    one = ingest.ingest_shard(torch.from_numpy(data).to(dev), cfg)
    assert torch.equal(read.read_store(manifest.merge_manifests([manifest.build_manifest(one)]), dev), torch.from_numpy(data).to(dev))


def test_global_l4_three_shards_equal_the_one_shard_run(dev):
    """Cross-shard base selection (SURVEY.md §8e last sentence, §8f-3), without a second GPU: three shards ingested with
    global_l4 — signatures 'all-gathered', one LSH over all stored chunks, remote base chunks fetched and appended as ghost
    chunks — give the streams, kinds and bases of the ONE-shard run, bit for bit; some dictionaries really are remote;
    all shards decode together (a DELTA's dictionary may be another shard's record) to the input; per-shard readers and
    the manifest packer refuse such a shard."""
    import torch
    from hmse_amd import IngestConfig, corpus, ingest, manifest, ops, read
    cfg = IngestConfig(seg_size=1 << 20)
    data = corpus.wiki_synth(12 << 20, seed=42)
    d = torch.from_numpy(data).to(dev)
    one = ingest.ingest_shard(d, cfg)
    shards = [d[: 4 << 20], d[4 << 20: 8 << 20], d[8 << 20:]]
    res = ingest.ingest_shards_local(shards, cfg, global_l4=True)
    assert torch.equal(torch.cat([r.streams for r in res]), one.streams)
    assert torch.equal(torch.cat([r.kind for r in res]), one.kind)
    assert torch.equal(torch.cat([r.base_global for r in res]), one.base)
    assert torch.equal(torch.cat([r.band_keys for r in res]), one.band_keys)
    n_remote = sum(int(((r.base_global >= 0) & (r.base < 0)).sum()) for r in res)
    n_remote_delta = sum(int(((r.base_global >= 0) & (r.base < 0) & (r.kind == 2)).sum()) for r in res)
    assert n_remote > 20 and n_remote_delta > 10
    back = read.reconstruct_shards(res, verify=True)
    assert torch.equal(torch.cat(back), d)
    # shard-local L4 (the default) stores more: its CF is what global L4 improves on
    loc = ingest.ingest_shards_local(shards, cfg)
    assert sum(int(r.streams.numel()) for r in loc) > int(one.streams.numel())
    assert torch.equal(torch.cat(read.reconstruct_shards(loc)), d)               # the joint reader also takes local-base shards
    late = next(r for r in res if bool(((r.base_global >= 0) & (r.base < 0)).any()))
    with pytest.raises(read.ReadError):
        read.reconstruct_shard(late)
    # the sharded STORE of such a run: per-shard manifests whose cross-shard DeltaChunk headers are packed unresolved and
    # listed in the manifest's remote_bases table; the merge fills them from the owning shard's index; the store reads back
    # on the GPU (one inflate over all shards) and through stock zlib on the host
    import manifest_ref
    parts = [manifest.build_manifest(r, i, 3) for i, r in enumerate(res)]
    for i, (r, m) in enumerate(zip(res, parts)):
        assert manifest_ref.build(_to_host(r), i, 3).to_bytes() == m.to_bytes()
    assert sum(m.n_remote() for m in parts) == n_remote_delta and parts[0].n_remote() == 0
    with pytest.raises(read.ReadError):
        read.read_store(manifest.Store(parts), dev)                               # unresolved pointers / headers
    with pytest.raises(read.ReadError):
        read.read_manifest(next(m for m in parts if m.n_remote()), dev)
    store = manifest.Store.from_bytes(manifest.merge_manifests(parts).to_bytes())
    assert sum(m.n_remote() for m in store.shards) == n_remote_delta
    assert torch.equal(read.read_store(store, dev, verify=True), d)
    assert manifest.reconstruct(store) == data.tobytes()


def test_distributed_ingest_world_size_1_runs_rccl(dev):
    """ingest_shard(distributed=True) under the driver: RCCL init, the count and digest all-gathers and the gathered dedupe
    run at world size 1 and give the single-shard result (the N > 1 exchange itself is covered by the gloo tests); with
    global_l4 also the signature all-gather and the base-fetch all-to-alls."""
    import os
    import torch
    import torch.distributed as dist
    from hmse_amd import IngestConfig, corpus, ingest, manifest, read
    cfg = IngestConfig(seg_size=1 << 20)
    data = torch.from_numpy(corpus.wiki_synth(3 << 20, seed=42)).to(dev)
    ref = ingest.ingest_shard(data, cfg)
    import socket
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        res = ingest.ingest_shard(data, cfg, distributed=True)
        glob = ingest.ingest_shard(data, cfg, distributed=True, global_l4=True)   # + signature all-gather and the three all-to-alls (empty at world size 1)
    finally:
        dist.destroy_process_group()
    assert torch.equal(glob.streams, ref.streams) and torch.equal(glob.base_global, ref.base) and torch.equal(glob.base, ref.base)
    assert res.shard_bases == [0] and res.chunk_base == 0 and res.n_global == ref.cuts.numel() - 1
    for a, b in ((res.first_occ, ref.first_occ), (res.uniq_ids, ref.uniq_ids), (res.base, ref.base), (res.streams, ref.streams), (res.kind, ref.kind)):
        assert torch.equal(a, b)
    assert torch.equal(read.read_store(manifest.merge_manifests([manifest.build_manifest(res, 0, 1)]), dev), data)


def test_document_aligned_two_shard_run_equals_oracle_with_the_same_segments(orc, dev):
    """north_star "the corpus shards naturally by document": segments built from document boundaries (hmse_amd.partition),
    whole documents dealt to two shards, every output equal to the oracle run over the same seg_off — and no chunk spans
    a document boundary that is a segment boundary."""
    import torch
    from hmse_amd import IngestConfig, corpus, ingest, partition
    cfg = IngestConfig(seg_size=256 << 10)
    oc = orc.default_cfg(**asdict(cfg))
    data = corpus.wiki_synth(5 << 20, seed=42)
    data[(3 << 20): (3 << 20) + 400000] = data[100000: 500000]                   # duplicates across the two shards
    t = torch.from_numpy(data).to(dev)
    so = partition.document_seg_off(partition.document_starts(t), data.size, cfg.seg_size)
    assert len(so) > 15 and (np.diff(so) <= cfg.seg_size).all() and (np.diff(so) % 4096 != 0).any()     # ragged segments
    shards, seg_offs, want_cuts, want_dg, bounds = [], [], [], [], []
    for a, b in partition.deal_segments(so, 2):
        lo, hi, rel = partition.shard_seg_off(so, a, b, dev)
        shards.append(t[lo:hi]); seg_offs.append(rel); bounds.append((lo, hi))
        c = orc.cdc(data[lo:hi], oc, rel.cpu().numpy().astype(np.uint64))
        want_cuts.append(c); want_dg.append(orc.sha256_chunks(data[lo:hi], c))
    res = ingest.ingest_shards_local(shards, cfg, seg_offs=seg_offs)
    fo, rc = orc.dedup(np.concatenate(want_dg))
    base0 = 0
    for r in range(2):
        lo, hi = bounds[r]
        n = len(want_cuts[r]) - 1
        assert np.array_equal(res[r].cuts.cpu().numpy().astype(np.uint64), want_cuts[r])
        assert np.isin(seg_offs[r].cpu().numpy()[1:], res[r].cuts.cpu().numpy()).all()     # every segment (document) boundary is a cut
        fol = fo[base0: base0 + n]
        assert np.array_equal(res[r].first_occ.cpu().numpy().astype(np.uint64), fol)
        uniq = np.nonzero(fol == np.arange(base0, base0 + n))[0].astype(np.uint64)
        sig = orc.minhash_chunks(data[lo:hi], want_cuts[r], oc, uniq)
        _, base = orc.lsh(sig, oc)
        out, off, kind = orc.deflate_chunks(data[lo:hi], want_cuts[r], oc, uniq, base)
        assert np.array_equal(res[r].uniq_ids.cpu().numpy().astype(np.uint64), uniq)
        assert np.array_equal(res[r].base.cpu().numpy(), base) and np.array_equal(res[r].kind.cpu().numpy(), kind)
        assert np.array_equal(res[r].streams.cpu().numpy(), out)
        base0 += n
    assert int((res[1].first_occ < res[1].chunk_base).sum()) > 20                # cross-shard POINTERs
