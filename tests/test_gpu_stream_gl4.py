"""GPU: global L4 of a multi-rank stream as hipGraph-captured phases (hmse_amd/stream_gl4.py; VERDICT r3 item 6, SURVEY.md §8f-3,
BASELINE configs[4]).  Bar: every rank's records, put in global stored order, are bit-identical to ONE pass of the oracle over the
logical stream — the same check the eager implementation (stream_dist.GlobalL4StreamIngest) passes — with every phase replayed from
a hipGraph; with one rank the captured chain equals one-shot ingest, also with the RCCL collectives between the replays."""
import numpy as np
import pytest

from test_gpu_stream_dist import NAMES, _dataset, oracle_whole_stream

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("world,graph", [(2, True), (3, True), (2, False)])
def test_captured_global_l4_stream_stores_the_bytes_of_the_one_rank_run(orc, dev, world, graph):
    import torch
    from hmse_amd import IngestConfig, manifest, read, stream_dist, stream_gl4
    cfg = IngestConfig(seg_size=1 << 20)
    data = _dataset().copy()
    B = 4 << 20
    # near-duplicates whose originals sit on a LATER-numbered rank of an earlier batch
    src = data[(2 << 20) + 300000: (2 << 20) + 300000 + 700000].copy()
    src[::1500] ^= 0x20
    data[(8 << 20) + 50000: (8 << 20) + 50000 + src.size] = src
    want = oracle_whole_stream(orc, data, cfg)
    batches = [torch.from_numpy(data[a: a + B].copy()) for a in range(0, data.size, B)]
    res = stream_gl4.stream_shards_local_gl4_graph(batches, cfg, world, dev, graph=graph)
    n_global = len(want["cuts"]) - 1
    assert all(r.n_global == n_global for r in res)
    U = len(want["uniq"])
    assert sorted(np.concatenate([r.ug.cpu().numpy() for r in res]).tolist()) == list(range(U))
    n_remote = n_later = 0
    for rank, r in enumerate(res):
        g = r.gidx.cpu().numpy()
        c = r.cuts.cpu().numpy().astype(np.int64)
        gl = np.diff(want["cuts"].astype(np.int64))
        assert np.array_equal(np.diff(c), gl[g])
        assert np.array_equal(r.first_occ.cpu().numpy().astype(np.uint64), want["first_occ"][g])
        assert np.array_equal(r.refcount.cpu().numpy().astype(want["refcount"].dtype), want["refcount"][g])
        assert np.array_equal(r.digests.cpu().numpy(), want["digests"][g])
        ug = r.ug.cpu().numpy()
        assert np.array_equal(g[r.uniq_ids.cpu().numpy()].astype(np.uint64), want["uniq"][ug])
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
    store = manifest.Store.from_bytes(manifest.merge_manifests(parts).to_bytes())
    assert manifest.reconstruct(store) == data.tobytes()
    assert np.array_equal(read.read_store(store, dev).cpu().numpy(), data)


def test_one_rank_captured_global_l4_chain_equals_one_shot_ingest_also_over_rccl(dev):
    """One rank: the global numbering IS the local one, so the four captured phases must give the one-shot result; run once plainly and
    once with always_exchange on backend "nccl" (world size 1): replay A -> RCCL all-gather -> replay B1 -> RCCL all-gather -> replay B2 ->
    replay B3 per batch — what one GPU can show of the N-rank loop on RCCL itself."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from hmse_amd import IngestConfig, ingest, stream_gl4
    cfg = IngestConfig(seg_size=1 << 20)
    data = _dataset()[: 10 << 20]
    whole = ingest.ingest_shard(torch.from_numpy(data).to(dev), cfg)
    P = 2 << 20

    def run(always):
        s = stream_gl4.GraphGlobalL4StreamIngest(cfg, data.size, P, dev, 1, 0, graph=True, always_exchange=always)
        for a in range(0, data.size, P):
            s.push(torch.from_numpy(data[a: a + P].copy()).pin_memory())
        res = s.finish()
        torch.cuda.synchronize()
        e = s._graphs[P]
        assert all(e[i] is not None for i in (1, 2, 3, 4)) and e[5] == 5
        return res
    res = run(False)
    for name in NAMES:
        assert torch.equal(getattr(res, name), getattr(whole, name)), name
    assert torch.equal(res.base_global, whole.base) and torch.equal(res.ug, torch.arange(res.ug.numel(), device=dev))
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        res2 = run(True)
    finally:
        dist.destroy_process_group()
    for name in NAMES:
        assert torch.equal(getattr(res2, name), getattr(whole, name)), name
