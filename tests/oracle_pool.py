"""The oracle pipeline of one shard with its two heavy stages (MinHash, DEFLATE) spread over worker processes.

TEST INFRASTRUCTURE (tests/ may use oracle/).  Workers are SPAWNED (the calling test process has the GPU initialised, so
it must not fork) and map the corpus from a file in /dev/shm.  The result equals test_gpu_ingest.oracle_pipeline's for
one shard — the same C oracle calls, only sliced: a DEFLATE slice carries the base chunks it needs as extra selection
entries, whose own output is dropped.
"""
from __future__ import annotations

import multiprocessing as mp
import os
import tempfile
from dataclasses import asdict

import numpy as np

_DATA = None


def _init(path, n):
    global _DATA
    _DATA = np.memmap(path, dtype=np.uint8, mode="r", shape=(n,))


def _cfg(kw):
    from oracle import oracle as O
    return O, O.default_cfg(**kw)


def _minhash(args):
    cuts, kw, ids = args
    O, oc = _cfg(kw)
    return O.minhash_chunks(_DATA, cuts, oc, ids)


def _deflate(args):
    cuts, kw, ids, base, take = args
    O, oc = _cfg(kw)
    out, off, kind = O.deflate_chunks(_DATA, cuts, oc, ids, base)
    return out[: int(off[take])].copy(), off[: take + 1].copy(), kind[:take].copy()


def pipeline(orc, data: np.ndarray, cfg, workers: int | None = None, slices_per_worker: int = 4) -> dict:
    workers = workers or min(os.cpu_count() or 1, 16)
    kw = asdict(cfg)
    oc = orc.default_cfg(**kw)
    cuts = orc.cdc(data, oc)
    dg = orc.sha256_chunks(data, cuts)
    fo, _ = orc.dedup(dg)
    uniq = np.nonzero(fo == np.arange(len(fo)))[0].astype(np.uint64)
    fd, path = tempfile.mkstemp(prefix="hmse_oracle_", suffix=".bin", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        with os.fdopen(fd, "wb") as f:
            f.write(np.ascontiguousarray(data).tobytes())
        with mp.get_context("spawn").Pool(workers, initializer=_init, initargs=(path, data.size)) as pool:
            nsl = max(1, min(len(uniq), workers * slices_per_worker))
            bounds = [len(uniq) * i // nsl for i in range(nsl + 1)]
            sig = np.concatenate(pool.map_async(_minhash, [(cuts, kw, uniq[a:b]) for a, b in zip(bounds[:-1], bounds[1:])]).get(timeout=900))
            _, base = orc.lsh(sig, oc)
            jobs = []
            for a, b in zip(bounds[:-1], bounds[1:]):
                bs = base[a:b]
                extra = np.unique(bs[(bs >= 0) & ((bs < a) | (bs >= b))])          # bases outside the slice
                ids = np.concatenate([uniq[a:b], uniq[extra]])
                remap = {int(e): (b - a) + i for i, e in enumerate(extra)}
                bl = np.array([-1 if v < 0 else (int(v) - a if a <= v < b else remap[int(v)]) for v in bs] + [-1] * len(extra), dtype=np.int64)
                jobs.append((cuts, kw, ids, bl, b - a))
            parts = pool.map_async(_deflate, jobs).get(timeout=900)   # a timeout: a worker that cannot start must not hang the run
    finally:
        os.unlink(path)
    out = np.concatenate([p[0] for p in parts]) if parts else np.zeros(0, np.uint8)
    off = [np.zeros(1, np.uint64)]
    run = np.uint64(0)
    for p in parts:
        off.append(p[1][1:] + run)
        run = run + p[1][-1]
    return dict(cuts=cuts, dg=dg, fo=fo, uniq=uniq, sig=sig, base=base, out=out, off=np.concatenate(off).astype(np.uint64),
                kind=np.concatenate([p[2] for p in parts]))
