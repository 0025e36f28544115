"""Host reference writer of the manifest records (numpy, vectorised): the CHECKER of the GPU packer (hmse_manifest_pack)
and the way CPU tests obtain a Manifest from oracle outputs.  TEST INFRASTRUCTURE — the product packs on the GPU
(hmse_amd/manifest.py build_manifest refuses host tensors).  Layouts: README.md:1263-1270, 2182-2189, 1312, 1448."""
import numpy as np

from hmse_amd.config import KIND_DELTA, KIND_POINTER
from hmse_amd.manifest import CHUNK_INDEX_DTYPE, DELTA_HDR_DTYPE, MAP_DTYPE, POINTER_DTYPE, PTR_UNRESOLVED, REMOTE_BASE_DTYPE, Manifest


def build(res, shard: int = 0, n_shards: int = 1) -> Manifest:
    cuts = res.cuts.cpu().numpy().astype(np.int64)
    n = len(cuts) - 1
    lens = np.diff(cuts)
    uniq = res.uniq_ids.cpu().numpy().astype(np.int64)
    u = len(uniq)
    off = res.stream_off.cpu().numpy().astype(np.int64)
    kind_u = res.kind.cpu().numpy()
    base = res.base.cpu().numpy() if res.base is not None else np.full(u, -1, np.int64)
    streams = res.streams.cpu().numpy()
    slen = np.diff(off)
    hdr = np.where(kind_u == KIND_DELTA, 8, 0)
    rec_len = slen + hdr
    total = int(rec_len.sum())
    unit = 1
    while (total + unit * u) // unit >= 2**32:
        unit *= 2
    rec_off = np.zeros(u + 1, np.int64)
    np.cumsum((rec_len + unit - 1) // unit * unit, out=rec_off[1:])
    blob = np.zeros(total if unit == 1 else int(rec_off[-1]), np.uint8)
    index = np.zeros(u, CHUNK_INDEX_DTYPE)
    index["lba"] = rec_off[:-1] // unit
    index["length"] = rec_len
    if res.digests is not None:
        index["sha256"] = res.digests.cpu().numpy()[uniq]
        index["refcount"] = np.minimum(res.refcount.cpu().numpy()[uniq], 65535)
    else:
        index["refcount"] = 1
    # stream bytes: destination of stream byte j of record k = rec_off[k] + hdr[k] + j
    if off[-1]:
        dst = np.repeat(rec_off[:-1] + hdr - off[:-1], slen) + np.arange(off[-1])
        blob[dst] = streams[: off[-1]]
    d = np.nonzero(kind_u == KIND_DELTA)[0]
    # global L4: a DELTA whose dictionary is another shard's record -> unresolved header + an entry of the remote table
    bg = res.base_global.cpu().numpy() if getattr(res, "base_global", None) is not None else None
    remote = None
    if bg is not None:
        rs = np.nonzero((bg >= 0) & (base < 0) & (kind_u == KIND_DELTA))[0]
        if len(rs):
            ub = np.asarray(res.u_bases, np.int64)
            sh = np.searchsorted(ub, bg[rs], side="right") - 1
            remote = np.zeros(len(rs), REMOTE_BASE_DTYPE)
            remote["slot"] = rs; remote["shard"] = sh; remote["base_slot"] = bg[rs] - ub[sh]
    if len(d):
        h = np.zeros(len(d), DELTA_HDR_DTYPE)
        loc = base[d] >= 0
        h["base_lba"] = np.where(loc, index["lba"][np.where(loc, base[d], 0)], 0xFFFFFFFF)
        h["base_length"] = np.where(loc, index["length"][np.where(loc, base[d], 0)], 0); h["delta_length"] = slen[d]
        pos = rec_off[d][:, None] + np.arange(8)[None, :]
        blob[pos] = np.frombuffer(h.tobytes(), np.uint8).reshape(len(d), 8)
    cmap = np.zeros(n, MAP_DTYPE)
    cmap["raw_length"] = np.minimum(lens, 65535)
    slot_of = np.full(n, -1, np.int64)
    slot_of[uniq] = np.arange(u)
    cb = int(res.chunk_base)
    g = res.first_occ.cpu().numpy().astype(np.int64) if res.first_occ is not None else np.arange(cb, cb + n)
    local = (g >= cb) & (g < cb + n)
    loc = np.where(local, g - cb, 0)
    is_ptr = g != np.arange(cb, cb + n)
    bases = np.asarray(res.shard_bases if res.shard_bases is not None else [0], np.int64)
    tgt_shard = np.where(local, shard, np.searchsorted(bases, g, side="right") - 1)
    cmap["slot"] = np.where(local, slot_of[loc], g - bases[tgt_shard])
    cmap["kind"] = np.where(is_ptr, KIND_POINTER, kind_u[np.where(local, slot_of[loc], 0)])
    cmap["shard"] = tgt_shard
    ptr = np.zeros(int(is_ptr.sum()), POINTER_DTYPE)
    pl = local[is_ptr]
    tslot = slot_of[loc[is_ptr]]
    ptr["target_lba"] = np.where(pl, index["lba"][np.where(pl, tslot, 0)] if u else 0, 0xFFFFFFFF)
    ptr["target_length"] = np.where(pl, index["length"][np.where(pl, tslot, 0)] if u else 0, 0)
    ptr["flags"] = KIND_POINTER | (tgt_shard[is_ptr] << 4) | np.where(pl, 0, PTR_UNRESOLVED)
    return Manifest(unit, index, cmap, ptr, blob, shard, n_shards, cb, remote)
