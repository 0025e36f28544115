"""CPU, world_size 2 over gloo: the one data-path collective (digest all-gather) and the shard-count
invariance of the dedupe rule.  The RCCL call itself only runs on the GPU box."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, digests_by_rank, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hmse_amd import ingest
    alld, base, n, bases = ingest.gather_digests(torch.from_numpy(digests_by_rank[rank]))
    out_q.put((rank, alld.numpy().copy(), base, n, bases))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_digests_and_global_dedup_two_ranks(orc):
    from hmse_amd import corpus
    data = corpus.wiki_synth(8 << 20, seed=42)
    # make the second shard repeat part of the first so that cross-shard duplicates exist
    data[6 << 20: 7 << 20] = data[1 << 20: 2 << 20]
    cfg = orc.default_cfg()
    seg = cfg.seg_size
    halves = [data[: 4 << 20], data[4 << 20:]]
    cuts = [orc.cdc(h, cfg) for h in halves]                          # shards are whole numbers of segments
    dg = [orc.sha256_chunks(h, c) for h, c in zip(halves, cuts)]
    # shard-count invariance of L2: the concatenated per-shard cuts equal the 1-shard cuts
    whole = orc.cdc(data, cfg)
    joined = np.concatenate([cuts[0], cuts[1][1:] + np.uint64(4 << 20)])
    assert np.array_equal(whole, joined) and seg == 4 << 20
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, dg, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref_all = np.concatenate(dg)
    for rank, alld, base, n, bases in got:
        assert n == len(ref_all) and base == (0 if rank == 0 else len(dg[0])) and bases == [0, len(dg[0])]
        assert np.array_equal(alld, ref_all)                           # (rank, local) order, identical on every rank
    fo_sharded, rc_sharded = orc.dedup(ref_all)
    fo_single, rc_single = orc.dedup(orc.sha256_chunks(data, whole))
    assert np.array_equal(fo_sharded, fo_single) and np.array_equal(rc_sharded, rc_single)
    assert (fo_sharded[len(dg[0]):] < len(dg[0])).sum() > 50          # cross-shard pointers exist
