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


def _fetch_worker(rank, world, port, shards, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hmse_amd import ingest
    data, cuts, uniq = (torch.from_numpy(x) for x in shards[rank][:3])
    req = torch.from_numpy(shards[rank][3])

    def gather(out_cuts, cid, cuts_, data_):      # the HIP gather (hmse_read_assemble) restated for the CPU test
        return torch.cat([data_[int(cuts_[c]): int(cuts_[c + 1])] for c in cid.tolist()]) if cid.numel() else data_[:0]
    u_bases = [0, len(shards[0][2]), len(shards[0][2]) + len(shards[1][2])]
    got, lens = ingest.fetch_chunks(req, u_bases, data, cuts, uniq, gather=gather)
    out_q.put((rank, got.numpy().copy(), lens.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_cross_rank_base_fetch_three_ranks():
    """The exchange behind ingest_shard(global_l4=True): every rank asks the owners for the stored chunks it needs as
    dictionaries (global stored-chunk ids, ascending) and gets their bytes back in request order — three all-to-alls,
    uneven splits, a rank that asks for nothing, a rank nobody asks."""
    rng = np.random.default_rng(11)
    shards, stored = [], []
    for r in range(3):
        lens = rng.integers(1, 4000, 40 + 7 * r)
        cuts = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        data = rng.integers(0, 256, int(cuts[-1]), dtype=np.uint8)
        uniq = np.sort(rng.choice(len(lens), 25 + r, replace=False)).astype(np.int64)     # the chunks this rank stores
        shards.append([data, cuts, uniq, None])
        stored += [data[cuts[c]: cuts[c + 1]] for c in uniq]
    ub = [0, 25, 51, 78]
    reqs = [np.array([30, 31, 60, 77], np.int64),           # rank 0 needs chunks of ranks 1 and 2
            np.zeros(0, np.int64),                           # rank 1 needs nothing
            np.array([0, 24, 25, 50], np.int64)]             # rank 2 needs chunks of ranks 0 and 1 (nobody asks rank 2's own last chunks twice)
    for r in range(3):
        shards[r][3] = reqs[r]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fetch_worker, args=(r, 3, port, shards, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(3)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, payload, lens in got:
        want = [stored[g] for g in reqs[rank]]
        assert lens.tolist() == [len(w) for w in want]
        assert np.array_equal(payload, np.concatenate(want) if want else np.zeros(0, np.uint8))
    assert ub[3] == len(stored)


def _row_worker(rank, world, port, rows_by_batch, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hmse_amd import stream_dist
    got = []
    out = None
    for rows in rows_by_batch:                    # one collective per global batch, the same buffer re-used (as the captured chain does)
        out = stream_dist.all_gather_rows(torch.from_numpy(rows[rank]), world, out=out)
        got.append(out.numpy().copy())
    out_q.put((rank, got))
    dist.barrier()
    dist.destroy_process_group()


def test_stream_row_exchange_and_global_order_two_ranks(orc):
    """The per-batch exchange of a multi-rank stream (hmse_amd/stream_dist.py): every rank contributes a FIXED-SIZE row
    {u64 count, 24 B pad, digests} per global batch, one all-gather delivers the rows in rank order to everybody; the global
    chunk order (batch, rank, local) read off the rows is the natural order of the logical stream, so the first-occurrence rule
    over it equals the rule over one ingest of the whole stream.  Includes a batch in which a rank has nothing."""
    from hmse_amd import corpus
    world, seg, B = 2, 1 << 20, 4 << 20
    data = corpus.wiki_synth(9 << 20, seed=42)
    data[(5 << 20) + 777: (6 << 20)] = data[777: (1 << 20)]                     # duplicates across batches and ranks
    data[(1 << 20): (1 << 20) + 300000] = data[(3 << 20): (3 << 20) + 300000]    # a LATER rank's piece holds the first occurrence
    cfg = orc.default_cfg(seg_size=seg)
    cap = (B // world) // cfg.min_size + (B // world) // seg + 2                 # hmse_stream_row_bytes' chunk capacity
    row_bytes = 32 + 32 * cap
    rows_by_batch, stream_digests = [], []
    for b0 in range(0, data.size, B):
        n = min(B, data.size - b0)
        s = -(-n // seg)
        bd = [min(n, (r * s // world) * seg) for r in range(world)] + [n]
        rows = []
        for r in range(world):
            piece = data[b0 + bd[r]: b0 + bd[r + 1]]
            row = np.zeros(row_bytes, np.uint8)
            if piece.size:
                dg = orc.sha256_chunks(piece, orc.cdc(piece, cfg))
                row[:8] = np.frombuffer(np.uint64(len(dg)).tobytes(), np.uint8)
                row[32: 32 + dg.size] = dg.reshape(-1)
                stream_digests.append(dg)
            rows.append(row)
        rows_by_batch.append(rows)
    assert int(np.frombuffer(rows_by_batch[-1][0][:8].tobytes(), np.uint64)[0]) == 0     # the ragged last batch: rank 0 has nothing
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_row_worker, args=(r, world, port, rows_by_batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    glob = []
    for rank, per_batch in got:
        mine = []
        for b, allrows in enumerate(per_batch):
            assert np.array_equal(allrows, np.concatenate(rows_by_batch[b]))   # rank order, identical on every rank
            for r in range(world):
                row = allrows[r * row_bytes: (r + 1) * row_bytes]
                c = int(np.frombuffer(row[:8].tobytes(), np.uint64)[0])
                mine.append(row[32: 32 + 32 * c].reshape(c, 32))
        glob.append(np.concatenate(mine))
    assert np.array_equal(glob[0], glob[1]) and np.array_equal(glob[0], np.concatenate(stream_digests))
    # (batch, rank, local) order == the natural order of the stream: same first occurrences as one ingest of the whole
    whole = orc.sha256_chunks(data, orc.cdc(data, cfg))
    assert np.array_equal(glob[0], whole)
    fo, rc = orc.dedup(glob[0])
    fo1, rc1 = orc.dedup(whole)
    assert np.array_equal(fo, fo1) and np.array_equal(rc, rc1) and (fo != np.arange(len(fo))).sum() > 100


def _agree_worker(rank, world, port, q):
    """Two ranks of a global-L4 stream in which rank 1's stage is refused (a capacity): BOTH must raise, at the same point."""
    import os
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from hmse_amd.stream_dist import GlobalL4StreamIngest
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = object.__new__(GlobalL4StreamIngest)          # the agreement logic only: no device state
    s.world, s.rank, s.group, s.failed = world, rank, None, False

    def stage(ok):
        if not ok:
            raise ValueError("stream index capacity exceeded (max_chunks)")
        return 7
    out = []
    out.append(s._guard(stage, True))                 # batch 1: both fine
    try:
        s._guard(stage, rank != 1)                    # batch 2: rank 1 is refused
        out.append("no error")
    except ValueError as e:
        out.append(str(e))
    dist.barrier()                                    # nobody is left behind in a collective: both ranks reach this
    dist.destroy_process_group()
    q.put((rank, out))


def test_a_rank_local_refusal_in_a_global_l4_stream_raises_on_every_rank():
    """GlobalL4StreamIngest's stages raise rank-local host errors (capacities) between collectives; `_guard` / `_agree`
    exchange an error word first, so that every rank raises the same error instead of one rank leaving and its peers blocking."""
    import multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_agree_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        assert got[r][0] == 7
        assert "rank 1: stream index capacity exceeded" in got[r][1] and "abandoned on every rank" in got[r][1]
