"""GPU: multi-rank streaming (BASELINE.json configs[4]; hmse_amd/stream_dist.py) on ONE GPU — the ranks run in lock step, the
exchange is the concatenation the all-gather would deliver — against the CPU oracle of the same definition:
global batches dealt to the ranks as contiguous segment runs, global chunk order (batch, rank, local), dedupe global over
that order, L4 bases and dictionaries scoped to the rank.  Plus the capacity paths of the device-count chain: a batch that
does not fit sets the sticky status and is dropped — no fault, no write out of bounds."""
from dataclasses import asdict

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NAMES = ("cuts", "digests", "first_occ", "refcount", "uniq_ids", "sig", "band_keys", "base", "kind", "stream_off", "streams")


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
    v = variants_dataset(a)
    return np.concatenate([a, v, corpus.wiki_synth((12 << 20) - a.size - v.size, seed=7), a[: (1 << 20) + 12345]])


def deal(batch_bytes, world, seg):
    """The dealing rule, restated: rank r gets segments [r S / R, (r + 1) S / R) of the batch's S segments."""
    s = -(-batch_bytes // seg)
    return [min(batch_bytes, (r * s // world) * seg) for r in range(world)] + [batch_bytes]


def oracle_stream_pipeline(orc, data, cfg, world, batch_bytes):
    """CPU restatement of a `world`-rank stream (README.md:1519-1580 against one index, sharded): per global batch every
    rank chunks and hashes its piece; the digests of all pieces join ONE index in (batch, rank, local) order; a chunk is stored
    by the rank holding its first occurrence; MinHash/LSH/dictionary DEFLATE per rank over ITS stored chunks in arrival order."""
    oc = orc.default_cfg(**asdict(cfg))
    seg = cfg.seg_size
    pieces = [[] for _ in range(world)]       # per rank: (lo, hi) byte ranges of the logical stream
    order = []                                # (rank, lo, hi) in global order
    for b0 in range(0, data.size, batch_bytes):
        n = min(batch_bytes, data.size - b0)
        bd = deal(n, world, seg)
        for r in range(world):
            if bd[r + 1] > bd[r]:
                pieces[r].append((b0 + bd[r], b0 + bd[r + 1])); order.append((r, b0 + bd[r], b0 + bd[r + 1]))
    local = [np.concatenate([data[a:b] for a, b in p]) if p else np.zeros(0, np.uint8) for p in pieces]
    cuts = [[np.uint64(0)] for _ in range(world)]
    gidx = [[] for _ in range(world)]
    dg_all, g, loc_off = [], 0, [0] * world
    for r, a, b in order:
        c = orc.cdc(data[a:b], oc)
        dg_all.append(orc.sha256_chunks(data[a:b], c))
        cuts[r] += list(c[1:] + np.uint64(loc_off[r]))
        gidx[r] += list(range(g, g + len(c) - 1))
        g += len(c) - 1; loc_off[r] += b - a
    dg_all = np.concatenate(dg_all)
    fo, rc = orc.dedup(dg_all)
    out = []
    for r in range(world):
        cr = np.array(cuts[r], np.uint64); gi = np.array(gidx[r], np.int64)
        uniq = np.nonzero(fo[gi] == gi.astype(np.uint64))[0].astype(np.uint64) if len(gi) else np.zeros(0, np.uint64)
        sig = orc.minhash_chunks(local[r], cr, oc, uniq)
        keys, base = orc.lsh(sig, oc)
        streams, off, kind = orc.deflate_chunks(local[r], cr, oc, uniq, base)
        out.append(dict(data=local[r], cuts=cr, gidx=gi, digests=dg_all[gi], first_occ=fo[gi], refcount=rc[gi], uniq_ids=uniq, sig=sig,
                        band_keys=keys, base=base, kind=kind, stream_off=off, streams=streams, n_global=g))
    return out


def _same(res, want, tag):
    for name in NAMES + ("gidx",):
        got = getattr(res, name).cpu().numpy()
        w = want[name]
        if name in ("sig", "band_keys"):
            got = got.view(np.uint32)
        assert np.array_equal(got.astype(w.dtype) if got.dtype != w.dtype else got, w), (tag, name)
    assert res.n_global == want["n_global"], tag


@pytest.mark.parametrize("world,graph", [(2, True), (3, True), (3, False)])
def test_multi_rank_stream_equals_oracle(orc, dev, world, graph):
    import torch
    from hmse_amd import IngestConfig, read, stream_dist
    cfg = IngestConfig(seg_size=1 << 20)
    data = _dataset().copy()
    # batch 1's first pieces repeat bytes of batch 0's LAST piece: first occurrences on another rank — one with a higher rank number
    data[(4 << 20) + 100000: (4 << 20) + 100000 + 1200000] = data[(2 << 20) + 50000: (2 << 20) + 50000 + 1200000]
    B = 4 << 20                                                    # 3 full batches + a ragged one (a rank gets nothing in it when world == 3)
    want = oracle_stream_pipeline(orc, data, cfg, world, B)
    batches = [torch.from_numpy(data[a: a + B].copy()) for a in range(0, data.size, B)]
    res = stream_dist.stream_shards_local(batches, cfg, world, dev, graph=graph)
    for r in range(world):
        _same(res[r], want[r], (world, graph, r))
    # pointers and dictionaries cross batches; first occurrences live on other ranks, earlier AND later ones
    n_other = 0
    for r in range(world):
        own = np.isin(want[r]["first_occ"], want[r]["gidx"].astype(np.uint64))
        n_other += int((~own).sum())
    assert sum(int((res[r].kind == 2).sum()) for r in range(world)) > 5
    assert n_other > 50
    # the read path over all ranks' records returns every rank's bytes
    back = read.reconstruct_shards(res, verify=True)
    for r in range(world):
        assert np.array_equal(back[r].cpu().numpy(), want[r]["data"]), r
    # ... and the STORE: one manifest per rank in the store numbering (first occurrences on any other shard, also later ones),
    # merged; the readers restore the STREAM order from the shards' pieces — host zlib verifier, whole-store GPU read, ranges
    from hmse_amd import manifest
    sr = stream_dist.store_results(res)
    parts = [manifest.Manifest.from_bytes(manifest.build_manifest(sr[r], r, world).to_bytes()) for r in range(world)]
    assert all(p.pieces is not None and len(p.pieces) >= 3 for p in parts)
    store = manifest.Store.from_bytes(manifest.merge_manifests(parts).to_bytes())
    fwd = sum(int((p.chunk_map["shard"][p.chunk_map["kind"] == 1] > p.shard).sum()) for p in store.shards)
    assert fwd > 10                                                  # POINTERs to LATER-numbered shards exist
    assert manifest.reconstruct(store) == data.tobytes()
    assert np.array_equal(read.read_store(store, dev).cpu().numpy(), data)
    rd = read.StoreReader(store, dev)
    rng = np.random.default_rng(3)
    off = rng.integers(0, data.size - 50000, 200); ln = rng.integers(1, 50000, 200)
    for o, n, g in zip(off, ln, rd.read_ranges(list(zip(off.tolist(), ln.tolist())))):
        assert np.array_equal(g.cpu().numpy(), data[o: o + n]), (o, n)


def test_one_batch_is_the_sharded_one_shot_ingest_and_one_rank_is_the_single_rank_stream(dev):
    import torch
    from hmse_amd import IngestConfig, ingest, stream, stream_dist
    cfg = IngestConfig(seg_size=1 << 20)
    data = _dataset()[: 9 << 20]
    # (a) ONE global batch over 2 ranks == ingest_shards_local over the same two contiguous segment runs
    bd = stream_dist.deal_batch(data.size, 2, cfg.seg_size)
    shards = [torch.from_numpy(data[bd[r]: bd[r + 1]]).to(dev) for r in range(2)]
    one_shot = ingest.ingest_shards_local(shards, cfg)
    res = stream_dist.stream_shards_local([torch.from_numpy(data)], cfg, 2, dev, graph=False)
    for r in range(2):
        for name in NAMES:
            assert torch.equal(getattr(res[r], name), getattr(one_shot[r], name)), (r, name)
        assert torch.equal(res[r].gidx, torch.arange(one_shot[r].chunk_base, one_shot[r].chunk_base + res[r].gidx.numel(), device=dev))
    # (b) ONE rank == the single-rank captured chain == one ingest of the whole stream
    whole = ingest.ingest_shard(torch.from_numpy(data).to(dev), cfg)
    B = 2 << 20
    batches = [torch.from_numpy(data[a: a + B].copy()) for a in range(0, data.size, B)]
    (r1,) = stream_dist.stream_shards_local(batches, cfg, 1, dev, graph=True)
    st = stream.StreamIngest(cfg, data.size, dev, graph=True)
    for b in batches:
        st.push(b)
    r0 = st.finish()
    for name in NAMES:
        assert torch.equal(getattr(r1, name), getattr(whole, name)), name
        assert torch.equal(getattr(r0, name), getattr(whole, name)), name


def test_graphs_are_captured_once_per_piece_size(dev):
    import torch
    from hmse_amd import IngestConfig, corpus, stream_dist
    cfg = IngestConfig(seg_size=1 << 20)
    data = corpus.wiki_synth(8 << 20, seed=3)
    s = stream_dist.DistStreamIngest(cfg, data.size, 2 << 20, dev, 1, 0, graph=True)
    for a in range(0, data.size, 2 << 20):
        s.push(torch.from_numpy(data[a: a + (2 << 20)].copy()).pin_memory())
    res = s.finish()
    e = s._graphs[2 << 20]
    assert e[1] is not None and e[2] is not None and e[3] == 4      # eager once, captured at the second use, replayed after
    assert int(res.cuts[-1]) == data.size and res.n_global == res.cuts.numel() - 1


@pytest.mark.parametrize("which", ["chunks", "stored", "streams", "global"])
def test_capacity_overflow_sets_the_sticky_status_and_drops_the_batch(dev, which):
    """The device-count chain has no host in the loop to refuse a batch: a batch that does not fit (chunk arrays, stored-chunk
    arrays, stream bytes, the global index of a multi-rank stream) must set state[7], be dropped as a whole, and turn every
    later batch into a no-op — never write out of bounds (ADVICE r2 stream_batch.hip:85)."""
    import torch
    from hmse_amd import IngestConfig, corpus, ingest, stream, stream_dist
    cfg = IngestConfig(seg_size=1 << 20)
    data = corpus.wiki_synth(6 << 20, seed=5)
    B = 2 << 20
    first = ingest.ingest_shard(torch.from_numpy(data[:B]).to(dev), cfg)
    n1, u1, s1 = first.cuts.numel() - 1, first.uniq_ids.numel(), first.streams.numel()
    kw = {"chunks": dict(max_chunks=n1 + 10), "stored": dict(max_chunks=4 * n1), "streams": dict(stream_capacity=s1 + 1000), "global": {}}[which]
    if which == "global":
        s = stream_dist.DistStreamIngest(cfg, data.size, B, dev, 1, 0, max_chunks_global=n1 + 10, graph=True)
    else:
        s = stream.StreamIngest(cfg, data.size, dev, graph=True, **kw)
        if which == "stored":
            s.max_unique = u1 + 10                                    # the chain takes the bound as an argument
    guard = {}
    if which != "global":                                             # canaries behind the arrays the dropped batch must not touch
        for nm in ("_cuts", "_uniq", "_kind", "_stream_off"):
            guard[nm] = getattr(s, nm).clone()
    for a in range(0, data.size, B):
        s.push(torch.from_numpy(data[a: a + B].copy()))
    with pytest.raises(ValueError, match="status"):
        s.finish()
    st = s._state.tolist()
    want_bit = {"chunks": 1, "stored": 2, "streams": 0x100, "global": 1}[which]
    assert st[7] & want_bit, hex(st[7])
    # the first batch is intact, the counters are frozen there, nothing of the failed batches was committed
    assert st[0] == B and st[1] == n1 and st[3] == u1 and st[5] == s1 and st[8] == n1
    if which != "global":
        assert torch.equal(s._cuts[: n1 + 1], first.cuts) and torch.equal(s._uniq[:u1], first.uniq_ids)
        assert torch.equal(s._streams[:s1], first.streams) and torch.equal(s._kind[:u1], first.kind)
        if which == "chunks":
            assert torch.equal(s._cuts[n1 + 1:], guard["_cuts"][n1 + 1:])          # not one cut of the dropped batch was written
        if which == "stored":
            assert torch.equal(s._uniq[s.max_unique:], guard["_uniq"][s.max_unique:])


def test_a_one_rank_state_block_whose_global_count_disagrees_is_refused(dev):
    """ADVICE r3 (include/hmse.h, ABI 2): with one rank the global chunk count state[8] must equal the local one state[1].  A caller
    that resumes a stream with the version-1 state layout (state[8] = 0) would overwrite digests and first occurrences from index 0
    on; the chain refuses the batch with sticky status bit 5 and commits nothing."""
    import torch
    from hmse_amd import IngestConfig, corpus, stream
    cfg = IngestConfig(seg_size=1 << 20)
    data = corpus.wiki_synth(6 << 20, seed=6)
    B = 2 << 20
    s = stream.StreamIngest(cfg, data.size, dev, graph=True)
    s.push(torch.from_numpy(data[:B].copy()))
    s.push(torch.from_numpy(data[B: 2 * B].copy()))                    # (a push processes the piece pushed before it)
    torch.cuda.synchronize()
    st0 = s._state.tolist()
    assert st0[7] == 0 and st0[8] == st0[1] > 0 and st0[0] == B
    digests = s._digests[: st0[1]].clone()
    s._state[8] = 0                                                     # what a version-1 caller's resume would have left there
    s.push(torch.from_numpy(data[2 * B:].copy()))                       # processes the second piece against the bad state block
    with pytest.raises(ValueError, match="status"):
        s.finish()
    st = s._state.tolist()
    assert st[7] & 32, hex(st[7])
    assert st[0] == B and st[1] == st0[1] and st[3] == st0[3] and st[5] == st0[5]      # counters frozen at the last good batch
    assert torch.equal(s._digests[: st0[1]], digests)                   # and nothing was overwritten


def test_a_piece_refused_by_push_poisons_the_stream_instead_of_raising_alone(dev):
    """DistStreamIngest.push() is a collective step; a rank whose piece is refused (here: larger than the nominal piece
    size) must not raise alone and leave its peers in the batch's all-gather.  The piece is refused through the chain (sticky
    status bit 6, an empty row), the rank keeps in step, and finish() raises on every rank (the all-reduce of the status)."""
    import torch
    from hmse_amd import IngestConfig, corpus, stream_dist
    cfg = IngestConfig(seg_size=1 << 20)
    data = corpus.wiki_synth(5 << 20, seed=8)
    B = 2 << 20
    s = stream_dist.DistStreamIngest(cfg, 8 << 20, B, dev, 1, 0, graph=True)
    s.push(torch.from_numpy(data[:B].copy()))
    s.push(torch.from_numpy(data[B: B + B + (1 << 20)].copy()))           # 3 MiB > the nominal 2 MiB: refused, not raised
    s.push(torch.from_numpy(data[:B].copy()))                             # later pieces are no-ops
    with pytest.raises(ValueError, match="refused by push"):
        s.finish()
    st = s._state.tolist()
    assert st[7] & 64 and st[0] == B                                      # the first piece is intact, nothing else was committed


def _gloo_rank(rank, world, port, pieces, cfg_kw, piece_bytes, out_q):
    """One rank of a 2-process stream on GPU 0: the product loop (DistStreamIngest.push -> phase A -> all-gather of the rows over
    gloo -> phase B) between real processes."""
    import os
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from hmse_amd import IngestConfig, stream_dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    cfg = IngestConfig(**cfg_kw)
    total = sum(p.size for p in pieces[rank])
    s = stream_dist.DistStreamIngest(cfg, max(total, 1), piece_bytes, dev, world, rank, graph=True)
    for p in pieces[rank]:
        s.push(torch.from_numpy(p))
    res = s.finish()
    out_q.put((rank, {nm: getattr(res, nm).cpu().numpy() for nm in NAMES + ("gidx",)}, res.n_global))
    dist.barrier()
    dist.destroy_process_group()


def test_two_processes_over_gloo_equal_the_lock_step_emulation(dev):
    """The N-rank control flow between REAL processes (two ranks sharing this GPU, exchange over gloo staged through host
    memory — everything except RCCL itself): every rank's result equals what the one-process lock-step emulation computes."""
    import socket
    import torch
    import torch.multiprocessing as mp
    from hmse_amd import IngestConfig, stream_dist
    cfg = IngestConfig(seg_size=1 << 20)
    data = _dataset()
    B, world = 4 << 20, 2
    batches = [torch.from_numpy(data[a: a + B].copy()) for a in range(0, data.size, B)]
    want = stream_dist.stream_shards_local(batches, cfg, world, dev, graph=False)
    want = [{nm: getattr(r, nm).cpu().numpy() for nm in NAMES + ("gidx",)} for r in want]
    pieces = [[], []]
    for b in batches:
        bd = stream_dist.deal_batch(b.numel(), world, cfg.seg_size)
        for r in range(world):
            pieces[r].append(b[bd[r]: bd[r + 1]].numpy().copy())
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    from dataclasses import asdict
    procs = [ctx.Process(target=_gloo_rank, args=(r, world, port, pieces, asdict(cfg), 2 << 20, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, arrs, n_global in got:
        for nm in NAMES + ("gidx",):
            assert np.array_equal(arrs[nm], want[rank][nm]), (rank, nm)
    assert got[0][2] == got[1][2] == sum(len(w["gidx"]) for w in want)


def test_world_size_1_stream_with_the_rccl_all_gather_between_the_graph_replays(dev):
    """What one GPU can show of the N-rank loop on RCCL itself: backend "nccl" at world size 1 with always_exchange — every
    batch is replay(graph A) -> all_gather_into_tensor on RCCL's stream ordering -> replay(graph B) reading the gathered rows —
    and the result is the single-rank stream's.  (Two ranks cannot share one device under RCCL; N > 1 is the gloo test above.)"""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from hmse_amd import IngestConfig, ingest, stream_dist
    cfg = IngestConfig(seg_size=1 << 20)
    data = _dataset()[: 10 << 20]
    whole = ingest.ingest_shard(torch.from_numpy(data).to(dev), cfg)
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        s = stream_dist.DistStreamIngest(cfg, data.size, 2 << 20, dev, 1, 0, graph=True, always_exchange=True)
        for a in range(0, data.size, 2 << 20):
            s.push(torch.from_numpy(data[a: a + (2 << 20)].copy()).pin_memory())
        res = s.finish()
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    e = s._graphs[2 << 20]
    assert e[1] is not None and e[2] is not None and e[3] == 5
    for name in NAMES:
        assert torch.equal(getattr(res, name), getattr(whole, name)), name


def test_status_poll_reports_a_dropped_batch_before_finish(dev):
    """StreamIngest(poll_status_every=1): the sticky status word travels to pinned host memory behind every batch without a wait;
    once such a copy has completed, the next push() raises — a long stream learns of a dropped batch early."""
    import torch
    from hmse_amd import IngestConfig, corpus, ingest, stream
    cfg = IngestConfig(seg_size=1 << 20)
    data = corpus.wiki_synth(8 << 20, seed=5)
    B = 2 << 20
    n1 = ingest.ingest_shard(torch.from_numpy(data[:B]).to(dev), cfg).cuts.numel() - 1
    s = stream.StreamIngest(cfg, data.size, dev, graph=True, max_chunks=n1 + 10, poll_status_every=1)
    s.push(torch.from_numpy(data[:B].copy()))
    s.push(torch.from_numpy(data[B: 2 * B].copy()))          # processes batch 1 (fits)
    s.push(torch.from_numpy(data[2 * B: 3 * B].copy()))      # processes batch 2: dropped on the device, status bit 0
    torch.cuda.synchronize()
    with pytest.raises(ValueError, match="polled after batch 2"):
        s.push(torch.from_numpy(data[3 * B:].copy()))
    ok = stream.StreamIngest(cfg, data.size, dev, graph=True, poll_status_every=1)
    for a in range(0, data.size, B):
        ok.push(torch.from_numpy(data[a: a + B].copy()))
    res = ok.finish()
    assert int(res.cuts[-1]) == data.size and not ok._polls or all(int(h.item()) == 0 for h, _, _ in ok._polls)


# ---------------------------------------------------------------------------------------------------------------- global L4
def oracle_whole_stream(orc, data, cfg):
    """With GLOBAL L4 a multi-rank stream selects bases over the global stored-chunk order (batch, rank, local) == the stream
    order, and dedupes over the same order: cuts, first occurrences, stored chunks, bases, kinds and streams are those of ONE
    pass of the oracle over the logical stream (pieces are whole segments, so the chunking is the same too)."""
    oc = orc.default_cfg(**asdict(cfg))
    cuts = orc.cdc(data, oc)
    dg = orc.sha256_chunks(data, cuts)
    fo, rc = orc.dedup(dg)
    uniq = np.nonzero(fo == np.arange(len(fo), dtype=np.uint64))[0].astype(np.uint64)
    sig = orc.minhash_chunks(data, cuts, oc, uniq)
    keys, base = orc.lsh(sig, oc)
    streams, off, kind = orc.deflate_chunks(data, cuts, oc, uniq, base)
    return dict(cuts=cuts, digests=dg, first_occ=fo, refcount=rc, uniq=uniq, sig=sig, keys=keys, base=base, streams=streams, off=off, kind=kind)


@pytest.mark.parametrize("world", [2, 3])
def test_global_l4_stream_stores_the_bytes_of_the_one_rank_run(orc, dev, world):
    """stream_dist.GlobalL4StreamIngest in lock step on one GPU (BASELINE configs[4] with global L4): every rank's
    records, put in global stored order, are bit-identical to the oracle's single pass over the logical stream — bases on other
    ranks (earlier AND later-numbered ones, of earlier batches and of the same batch) included; the in-memory reader and the
    merged STORE (dictionaries on later-numbered shards: dependency-ordered decode) return the stream."""
    import torch
    from hmse_amd import IngestConfig, manifest, read, stream_dist
    cfg = IngestConfig(seg_size=1 << 20)
    data = _dataset().copy()
    B = 4 << 20
    # near-duplicates whose originals sit on a LATER-numbered rank of an earlier batch: batch 2's first piece (rank 0) gets lightly
    # edited copies of bytes of batch 0's last piece
    src = data[(2 << 20) + 300000: (2 << 20) + 300000 + 700000].copy()
    src[::1500] ^= 0x20
    data[(8 << 20) + 50000: (8 << 20) + 50000 + src.size] = src
    want = oracle_whole_stream(orc, data, cfg)
    batches = [torch.from_numpy(data[a: a + B].copy()) for a in range(0, data.size, B)]
    res = stream_dist.stream_shards_local_global_l4(batches, cfg, world, dev)
    n_global = len(want["cuts"]) - 1
    assert all(r.n_global == n_global for r in res)
    U = len(want["uniq"])
    assert sorted(np.concatenate([r.ug.cpu().numpy() for r in res]).tolist()) == list(range(U))
    n_remote = n_later = 0
    for rank, r in enumerate(res):
        g = r.gidx.cpu().numpy()
        c = r.cuts.cpu().numpy().astype(np.int64)
        gl = np.diff(want["cuts"].astype(np.int64))
        assert np.array_equal(np.diff(c), gl[g])                                               # this rank's chunks are the stream's
        assert np.array_equal(r.first_occ.cpu().numpy().astype(np.uint64), want["first_occ"][g])
        assert np.array_equal(r.refcount.cpu().numpy().astype(want["refcount"].dtype), want["refcount"][g])
        assert np.array_equal(r.digests.cpu().numpy(), want["digests"][g])
        ug = r.ug.cpu().numpy()
        assert np.array_equal(g[r.uniq_ids.cpu().numpy()].astype(np.uint64), want["uniq"][ug])  # stored chunk <-> global stored index
        assert np.array_equal(r.sig.cpu().numpy().view(np.uint32), want["sig"][ug])
        assert np.array_equal(r.band_keys.cpu().numpy().view(np.uint32), want["keys"][ug])
        assert np.array_equal(r.base_global.cpu().numpy(), want["base"][ug].astype(np.int64))
        assert np.array_equal(r.kind.cpu().numpy(), want["kind"][ug])
        so, st = r.stream_off.cpu().numpy(), r.streams.cpu().numpy()
        wo = want["off"].astype(np.int64)
        for j, u in enumerate(ug):
            assert np.array_equal(st[so[j]: so[j + 1]], want["streams"][wo[u]: wo[u + 1]]), (rank, j)
        if r.remote_bases is not None:
            n_remote += len(r.remote_bases)
            n_later += int((r.remote_bases["shard"] > rank).sum())
    assert n_remote > 10 and n_later > 5
    back = read.reconstruct_shards(res, verify=True)
    bounds = [stream_dist.deal_batch(b.numel(), world, cfg.seg_size) for b in batches]
    for r in range(world):
        mine = np.concatenate([b.numpy()[bd[r]: bd[r + 1]] for b, bd in zip(batches, bounds)])
        assert np.array_equal(back[r].cpu().numpy(), mine), r
    sr = stream_dist.store_results(res)
    parts = [manifest.Manifest.from_bytes(manifest.build_manifest(sr[r], r, world).to_bytes()) for r in range(world)]
    assert sum(p.n_remote() for p in parts) == n_remote
    store = manifest.Store.from_bytes(manifest.merge_manifests(parts).to_bytes())
    assert manifest.reconstruct(store) == data.tobytes()
    assert np.array_equal(read.read_store(store, dev).cpu().numpy(), data)
    rd = read.StoreReader(store, dev)
    rng = np.random.default_rng(11)
    off = rng.integers(0, data.size - 40000, 150); ln = rng.integers(1, 40000, 150)
    for o, n, gt in zip(off, ln, rd.read_ranges(list(zip(off.tolist(), ln.tolist())))):
        assert np.array_equal(gt.cpu().numpy(), data[o: o + n]), (o, n)


def _gloo_rank_global_l4(rank, world, port, pieces, cfg_kw, piece_bytes, out_q):
    import os
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from hmse_amd import IngestConfig, stream_dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    cfg = IngestConfig(**cfg_kw)
    s = stream_dist.GlobalL4StreamIngest(cfg, max(sum(p.size for p in pieces[rank]), 1), piece_bytes, dev, world, rank)
    for p in pieces[rank]:
        s.push(torch.from_numpy(p))
    res = s.finish()
    out_q.put((rank, {nm: getattr(res, nm).cpu().numpy() for nm in ("ug", "gidx", "base_global", "kind", "stream_off", "streams")}, s.remote_dictionaries))
    dist.barrier()
    dist.destroy_process_group()


def test_global_l4_stream_two_processes_over_gloo_equal_the_emulation(dev):
    """The product loop (push -> stage_hash -> all-gather digests -> stage_index -> all-gather signatures -> stage_lsh -> three
    all-to-alls for the remote dictionaries -> stage_encode) between two REAL processes sharing this GPU (gloo): equals lock step."""
    import socket
    import torch
    import torch.multiprocessing as mp
    from hmse_amd import IngestConfig, stream_dist
    cfg = IngestConfig(seg_size=1 << 20)
    data = _dataset()
    B, world = 4 << 20, 2
    batches = [torch.from_numpy(data[a: a + B].copy()) for a in range(0, data.size, B)]
    want = stream_dist.stream_shards_local_global_l4(batches, cfg, world, dev)
    names = ("ug", "gidx", "base_global", "kind", "stream_off", "streams")
    want = [{nm: getattr(r, nm).cpu().numpy() for nm in names} for r in want]
    pieces = [[], []]
    for b in batches:
        bd = stream_dist.deal_batch(b.numel(), world, cfg.seg_size)
        for r in range(world):
            pieces[r].append(b[bd[r]: bd[r + 1]].numpy().copy())
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_rank_global_l4, args=(r, world, port, pieces, asdict(cfg), 2 << 20, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, arrs, n_remote in got:
        for nm in names:
            assert np.array_equal(arrs[nm], want[rank][nm]), (rank, nm)
    assert sum(g[2] for g in got) > 5


def test_global_l4_stream_on_one_rank_is_the_one_shot_ingest(dev):
    """GlobalL4StreamIngest at world size 1 (no process group): push/finish like the other front ends, equal to one ingest."""
    import torch
    from hmse_amd import IngestConfig, ingest, stream_dist
    cfg = IngestConfig(seg_size=1 << 20)
    data = _dataset()[: 9 << 20]
    whole = ingest.ingest_shard(torch.from_numpy(data).to(dev), cfg)
    s = stream_dist.GlobalL4StreamIngest(cfg, data.size, 2 << 20, dev, 1, 0)
    for a in range(0, data.size, 2 << 20):
        s.push(torch.from_numpy(data[a: a + (2 << 20)].copy()))
    res = s.finish()
    for name in NAMES:
        assert torch.equal(getattr(res, name), getattr(whole, name)), name
    assert torch.equal(res.base_global, whole.base) and res.remote_bases is None and s.remote_dictionaries == 0


def test_chain_refuses_a_workspace_that_was_never_initialised(dev, monkeypatch):
    """The chain's MinHash memo table persists across batches, so the workspace must be prepared once (hmse_stream_workspace_init,
    ops.stream_workspace).  A chain handed a raw allocation finds no tag, drops the batch and sets sticky status bit 4 — never
    signatures computed from garbage table entries."""
    import torch
    from hmse_amd import IngestConfig, corpus, ops, stream
    cfg = IngestConfig(seg_size=1 << 20)
    data = corpus.wiki_synth(4 << 20, seed=5)
    monkeypatch.setattr(ops, "stream_workspace", lambda n, c, d: torch.zeros(ops.stream_batch_workspace_bytes(n, c), dtype=torch.uint8, device=d))
    s = stream.StreamIngest(cfg, data.size, dev, graph=True)
    for a in range(0, data.size, 2 << 20):
        s.push(torch.from_numpy(data[a: a + (2 << 20)].copy()))
    with pytest.raises(ValueError, match="status 0x10"):
        s.finish()
    assert s._state.tolist()[1] == 0 and s._state.tolist()[3] == 0       # nothing was committed
