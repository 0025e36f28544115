"""Stage operators L2/L3/L4/L1 over the C-ABI (include/hmse.h): tensors in, tensors out.

PyTorch is plumbing here (device memory, current HIP stream); every byte of work happens in the
hand-written gfx950 kernels of hmse_amd/csrc.  u64 arrays travel as torch.int64, u32 as torch.int32
(same bits).  All ops run on the tensor's device and torch's current stream and raise HmseError
on a non-zero status — there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .config import IngestConfig

STAGE_L2, STAGE_SHA, STAGE_DEDUP, STAGE_MINHASH, STAGE_LSH, STAGE_DEFLATE = 2, 3, 4, 5, 6, 7
STAGE_INFLATE, STAGE_ASSEMBLE, STAGE_MANIFEST = 16, 17, 18


class HmseError(RuntimeError):
    def __init__(self, code: int, where: str):
        msg = _lib.hip_lib().hmse_strerror(code).decode()
        super().__init__(f"{where}: {msg} ({code})")
        self.code = code


def _check(rc: int, where: str) -> None:
    if rc != 0:
        raise HmseError(rc, where)


def _require_gpu(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise HmseError(-1, f"{name} must live in HBM (got a {t.device} tensor; hmse_amd has no CPU path)")
    if not t.is_contiguous():
        raise HmseError(-1, f"{name} must be contiguous")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t) -> int | None:
    return None if t is None else t.data_ptr()


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def workspace_bytes(stage: int, n: int, cfg: IngestConfig) -> int:
    c = cfg.to_c()
    return int(_lib.hip_lib().hmse_workspace_bytes(stage, n, C.byref(c)))


def segment_offsets(n: int, seg_size: int, device) -> torch.Tensor:
    k = max(1, -(-n // seg_size))
    off = torch.arange(k + 1, dtype=torch.int64, device=device) * seg_size
    return torch.clamp(off, max=n)


def l2_cdc(data: torch.Tensor, cfg: IngestConfig, seg_off: torch.Tensor | None = None) -> torch.Tensor:
    """FastCDC cut points. Returns int64 cuts[n_chunks+1] (cuts[0] == 0). README.md:2456-2490."""
    _require_gpu(data, "data")
    n = data.numel()
    dev = data.device
    if seg_off is None:
        seg_off = segment_offsets(n, cfg.seg_size, dev)
    _require_gpu(seg_off, "seg_off")
    n_seg = seg_off.numel() - 1
    c = cfg.to_c()
    lib = _lib.hip_lib()
    cap = n // cfg.min_size + n_seg + 2
    cuts = torch.empty(cap, dtype=torch.int64, device=dev)
    meta = torch.zeros(2, dtype=torch.int64, device=dev)  # [n_cuts, status]
    # exact workspace for this segmentation: the generic sizing assumes ceil(n/seg_size) segments
    ws_bytes = workspace_bytes(STAGE_L2, n, cfg) + 24 * max(0, n_seg - n // cfg.seg_size) + 4096
    ws = _ws(ws_bytes, dev)
    rc = lib.hmse_l2_cdc(_ptr(data), n, _ptr(seg_off), n_seg, C.byref(c), _ptr(cuts), cap, meta.data_ptr(),
                         meta.data_ptr() + 8, ws.data_ptr(), ws.numel(), _stream())
    _check(rc, "hmse_l2_cdc")
    n_cuts, status = (int(v) for v in meta.tolist())
    if status & 0xFFFFFFFF:
        raise HmseError(-2, f"hmse_l2_cdc device status {status & 0xFFFFFFFF:#x}")
    return cuts[: n_cuts + 1]


def l3_sha256(data: torch.Tensor, cuts: torch.Tensor) -> torch.Tensor:
    """SHA-256 of every chunk -> uint8 [n_chunks, 32]. README.md:2543."""
    _require_gpu(data, "data")
    _require_gpu(cuts, "cuts")
    n_chunks = cuts.numel() - 1
    out = torch.empty((max(n_chunks, 0), 32), dtype=torch.uint8, device=data.device)
    if n_chunks <= 0:
        return out
    ws = _ws(workspace_bytes(STAGE_SHA, n_chunks, IngestConfig()), data.device)
    rc = _lib.hip_lib().hmse_l3_sha256(_ptr(data), data.numel(), _ptr(cuts), n_chunks, _ptr(out), ws.data_ptr(), ws.numel(), _stream())
    _check(rc, "hmse_l3_sha256")
    return out


def l3_dedup(digests: torch.Tensor):
    """first_occ int64[n], refcount int32[n]. README.md:1288-1292."""
    _require_gpu(digests, "digests")
    n = digests.shape[0]
    dev = digests.device
    fo = torch.empty(n, dtype=torch.int64, device=dev)
    rc_t = torch.empty(n, dtype=torch.int32, device=dev)
    if n == 0:
        return fo, rc_t
    c = IngestConfig().to_c()
    nb = int(_lib.hip_lib().hmse_workspace_bytes(STAGE_DEDUP, n, C.byref(c)))
    ws = _ws(nb, dev)
    rc = _lib.hip_lib().hmse_l3_dedup(_ptr(digests), n, _ptr(fo), _ptr(rc_t), ws.data_ptr(), ws.numel(), _stream())
    _check(rc, "hmse_l3_dedup")
    return fo, rc_t


def l3_index_slots(capacity_chunks: int) -> int:
    return int(_lib.hip_lib().hmse_l3_index_slots(int(capacity_chunks)))


def l3_index_update(digests_all: torch.Tensor, n_old: int, n_new: int, first_occ: torch.Tensor, refcount: torch.Tensor,
                    table: torch.Tensor) -> None:
    """Persistent L3 index: digests [n_old, n_old + n_new) join `table` (int32[l3_index_slots(capacity)], kept by the
    caller across calls); first_occ / refcount tails are written in place.  README.md:1288-1292."""
    for t, nm in ((digests_all, "digests"), (first_occ, "first_occ"), (refcount, "refcount"), (table, "table")):
        _require_gpu(t, nm)
    if digests_all.shape[0] < n_old + n_new or first_occ.numel() < n_old + n_new or refcount.numel() < n_old + n_new:
        raise HmseError(-1, "l3_index_update: arrays shorter than n_old + n_new")
    rc = _lib.hip_lib().hmse_l3_index_update(_ptr(digests_all), n_old, n_new, _ptr(first_occ), _ptr(refcount), _ptr(table), table.numel(), _stream())
    _check(rc, "hmse_l3_index_update")


def l4_lsh_slots(capacity_chunks: int) -> int:
    return int(_lib.hip_lib().hmse_l4_lsh_slots(int(capacity_chunks)))


def l4_lsh_update(sig_all: torch.Tensor, n_old: int, n_new: int, cfg: IngestConfig, band_keys: torch.Tensor, base: torch.Tensor | None,
                  tables: torch.Tensor, keys_given: bool = False) -> None:
    """Persistent L4b band tables: signatures [n_old, n_old + n_new) join `tables` (int32[bands, l4_lsh_slots(capacity)]);
    band_keys / base tails are written in place.  README.md:1554-1576, 1937-1945."""
    for t, nm in ((sig_all, "sig"), (band_keys, "band_keys"), (tables, "tables")):
        _require_gpu(t, nm)
    if sig_all.shape[0] < n_old + n_new or band_keys.shape[0] < n_old + n_new or (base is not None and base.numel() < n_old + n_new):
        raise HmseError(-1, "l4_lsh_update: arrays shorter than n_old + n_new")
    c = cfg.to_c()
    rc = _lib.hip_lib().hmse_l4_lsh_update(_ptr(sig_all), n_old, n_new, C.byref(c), _ptr(band_keys), _ptr(base), _ptr(tables), tables.shape[1],
                                          1 if keys_given else 0, _stream())
    _check(rc, "hmse_l4_lsh_update")


def l4_minhash(data: torch.Tensor, cuts: torch.Tensor, cfg: IngestConfig, chunk_ids: torch.Tensor | None = None,
               memo: bool = True) -> torch.Tensor:
    """MinHash signatures -> int32 [n_sel, 128] (uint32 bits). README.md:2578-2598.
    memo=False hands the C-ABI a workspace too small for its memo table (include/hmse.h): every hash is then computed."""
    _require_gpu(data, "data")
    _require_gpu(cuts, "cuts")
    if chunk_ids is not None:
        _require_gpu(chunk_ids, "chunk_ids")
    n_sel = (cuts.numel() - 1) if chunk_ids is None else chunk_ids.numel()
    sig = torch.empty((max(n_sel, 0), cfg.n_hashes), dtype=torch.int32, device=data.device)
    if n_sel <= 0:
        return sig
    c = cfg.to_c()
    ws = _ws(workspace_bytes(STAGE_MINHASH, n_sel, cfg) if memo else 256, data.device)
    rc = _lib.hip_lib().hmse_l4_minhash(_ptr(data), data.numel(), _ptr(cuts), _ptr(chunk_ids), n_sel, C.byref(c), _ptr(sig),
                                        ws.data_ptr(), ws.numel(), _stream())
    _check(rc, "hmse_l4_minhash")
    return sig


def l4_lsh(sig: torch.Tensor, cfg: IngestConfig):
    """band keys int32 [n, bands], base int64 [n] (-1 = none). README.md:1375-1383, 1987-1996."""
    _require_gpu(sig, "sig")
    n = sig.shape[0]
    dev = sig.device
    keys = torch.empty((n, cfg.bands), dtype=torch.int32, device=dev)
    base = torch.empty(n, dtype=torch.int64, device=dev)
    if n == 0:
        return keys, base
    c = cfg.to_c()
    nb = int(_lib.hip_lib().hmse_workspace_bytes(STAGE_LSH, n, C.byref(c)))
    ws = _ws(nb, dev)
    rc = _lib.hip_lib().hmse_l4_lsh(_ptr(sig), n, C.byref(c), _ptr(keys), _ptr(base), ws.data_ptr(), ws.numel(), _stream())
    _check(rc, "hmse_l4_lsh")
    return keys, base


def record_bytes(lens: torch.Tensor, has_dict=None) -> torch.Tensor:
    """hmse_l1_deflate_record_bytes() — where `has_dict` (bool tensor) is set: hmse_l1_deflate_record_bytes_dict() — for a
    tensor of chunk lengths (0 above 32768: such a chunk is never encoded)."""
    body = 1296 + 4 * ((lens + 3) & ~3)
    if has_dict is not None:
        body = body + torch.where(has_dict, lens + 21, torch.zeros_like(lens))
    return torch.where(lens <= 32768, (body + 255) & ~255, torch.zeros_like(lens))


DEFLATE_WS_LIMIT = 64 << 30   # upper bound of the per-job records one hmse_l1_deflate call may hold (l1_deflate splits a larger selection)


def deflate_ws_limit(device) -> int:
    """Record workspace one hmse_l1_deflate call may take: at most DEFLATE_WS_LIMIT and at most half of the HBM that is free
    right now (cached blocks of the allocator count as free) — a resident corpus, streams and index then still fit beside it."""
    free, _ = torch.cuda.mem_get_info(device)
    cached = torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
    return max(256 << 20, min(DEFLATE_WS_LIMIT, (free + cached) // 2))


# windows T = dictionary + chunk of the match kernel's size classes S, S2, SG, SG2, SG3 (hmse_amd/csrc/l1_deflate.hip: HMSE_TCAP_*; above: class B) —
# diagnostics only (bench.py's per-class algorithmic bytes, tools/): the library classifies on the device
DEFLATE_CLASS_CAPS = (10048, 13952, 17408, 22976, 32768)


def l1_deflate(data: torch.Tensor, cuts: torch.Tensor, cfg: IngestConfig, chunk_ids: torch.Tensor | None = None,
               base: torch.Tensor | None = None, base_is_chunk_id: bool = False, ws: torch.Tensor | None = None,
               ws_limit: int | None = None):
    """Per-chunk raw DEFLATE with the base chunk as dictionary (`base`: index into the selection, or — with
    base_is_chunk_id — a chunk index into `cuts`, e.g. a chunk stored by an earlier batch of a stream).

    The C-ABI call keeps one record per chunk (histograms, token list sized for the all-literal worst case — the FULL stream
    later overwrites it; a chunk with a dictionary has the DELTA stream's slot behind it: ~4.2 / 5.2 x the chunk) in its workspace.  A selection whose records exceed `ws_limit`
    bytes (default DEFLATE_WS_LIMIT) is encoded in consecutive pieces that each fit — same streams, same order, a bounded
    workspace whatever the shard size.

    Returns (out uint8[total], out_off int64[n_sel+1], kind uint8[n_sel]). README.md:2374-2378, 2182-2189."""
    if ws is not None:
        _require_gpu(ws, "ws")
    _require_gpu(data, "data")
    _require_gpu(cuts, "cuts")
    dev = data.device
    n_sel = (cuts.numel() - 1) if chunk_ids is None else chunk_ids.numel()
    out_off = torch.zeros(max(n_sel, 0) + 1, dtype=torch.int64, device=dev)
    kind = torch.zeros(max(n_sel, 0), dtype=torch.uint8, device=dev)
    if n_sel <= 0:
        return torch.empty(0, dtype=torch.uint8, device=dev), out_off, kind
    lens = (cuts[1:] - cuts[:-1]) if chunk_ids is None else (cuts[chunk_ids + 1] - cuts[chunk_ids])
    # per-chunk record of the C-ABI workspace = hmse_l1_deflate_record_bytes[_dict](len), evaluated on the device;
    # tests/test_abi.py holds the formulas together
    rec = record_bytes(lens, None if base is None else base >= 0)
    raw, need = (int(v) for v in torch.stack([lens.sum(), rec.sum()]).tolist())
    cap = raw + 5 * n_sel + 64  # a stored block is the worst case
    out = torch.empty(cap, dtype=torch.uint8, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    c = cfg.to_c()
    limit = deflate_ws_limit(dev) if ws_limit is None else int(ws_limit)
    if need <= limit:
        pieces = [(0, n_sel, need)]
        ids_all, base_all, by_id = chunk_ids, base, base_is_chunk_id
    else:
        # consecutive pieces of the selection whose records fit; dictionaries are named by chunk id so that a piece may
        # use a chunk of an earlier piece
        csum = torch.cumsum(rec, 0)
        ends, lo, start = [], 0, 0
        while start < n_sel:
            e = int(torch.searchsorted(csum, torch.tensor([lo + limit], dtype=csum.dtype, device=dev), right=True).item())
            e = max(e, start + 1)
            ends.append(e); lo = int(csum[e - 1].item()); start = e
        cs = [0] + [int(csum[e - 1].item()) for e in ends]
        pieces = [(a, e, cs[i + 1] - cs[i]) for i, (a, e) in enumerate(zip([0] + ends[:-1], ends))]
        ids_all = chunk_ids if chunk_ids is not None else torch.arange(n_sel, dtype=torch.int64, device=dev)
        base_all = None if base is None else (base if base_is_chunk_id else torch.where(base >= 0, ids_all[base.clamp(min=0)], base))
        by_id = True
    total = 0
    for a, e, need_p in pieces:
        n_p = e - a
        nb = int(_lib.hip_lib().hmse_workspace_bytes(STAGE_DEFLATE, n_p, C.byref(c)))
        if ws is None or ws.numel() < nb + need_p + 4096:   # (`ws`: a caller-held workspace, e.g. the streaming front end's, reused when big enough)
            ws = None
            ws = _ws(nb + need_p + 4096, dev)
        ids_p = None if ids_all is None else ids_all[a:e]
        base_p = None if base_all is None else base_all[a:e]
        off_p = out_off[a: e + 1] if len(pieces) == 1 else torch.zeros(n_p + 1, dtype=torch.int64, device=dev)
        rc = _lib.hip_lib().hmse_l1_deflate_ex(_ptr(data), data.numel(), _ptr(cuts), _ptr(ids_p), _ptr(base_p), n_p, C.byref(c),
                                               1 if by_id else 0, out.data_ptr() + total, cap - total, _ptr(off_p), _ptr(kind[a:e]), _ptr(status),
                                               ws.data_ptr(), ws.numel(), _stream())
        _check(rc, "hmse_l1_deflate")
        st, tot_p = (int(v) for v in torch.stack([status[0].to(torch.int64), off_p[-1]]).tolist())
        if st:
            raise HmseError(-2, f"hmse_l1_deflate device status {st:#x}")
        if len(pieces) > 1:
            out_off[a + 1: e + 1] = off_p[1:] + total
        total += tot_p
    return out[:total], out_off, kind


def l1_inflate_mode(mode: int) -> None:
    """0: pick the decoder by stream count (default); 1: one stream per wavefront; 2: one stream per lane (include/hmse.h)."""
    _check(_lib.hip_lib().hmse_l1_inflate_mode(int(mode)), "hmse_l1_inflate_mode")


def l1_inflate(streams: torch.Tensor, stream_off: torch.Tensor, kind: torch.Tensor, base: torch.Tensor | None,
               raw_len: torch.Tensor, stream_len: torch.Tensor | None = None, check: bool = True):
    """Raw-DEFLATE decode of stored chunks; DELTA chunks use their base chunk's raw bytes as dictionary.

    `stream_off` int64[n+1] (dense) or int64[n] starts with `stream_len` int32[n]; `raw_len` int64[n] raw chunk sizes.
    Returns (raw uint8[sum(raw_len)], raw_off int64[n+1], ok uint8[n]); with check=True a corrupt record raises.
    README.md:2397-2400, 1635-1669."""
    _require_gpu(streams, "streams")
    _require_gpu(stream_off, "stream_off")
    _require_gpu(kind, "kind")
    _require_gpu(raw_len, "raw_len")
    dev = streams.device
    n_sel = kind.numel()
    if stream_off.numel() != (n_sel if stream_len is not None else n_sel + 1) or raw_len.numel() != n_sel:
        raise HmseError(-1, "l1_inflate: stream_off / raw_len do not match kind")
    raw_off = torch.zeros(n_sel + 1, dtype=torch.int64, device=dev)
    torch.cumsum(raw_len, 0, out=raw_off[1:])
    total = int(raw_off[-1].item()) if n_sel else 0
    raw = torch.empty(total, dtype=torch.uint8, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    ok = torch.zeros(n_sel, dtype=torch.uint8, device=dev)
    if n_sel == 0:
        return raw, raw_off, ok
    if base is not None:
        _require_gpu(base, "base")
    if stream_len is not None:
        _require_gpu(stream_len, "stream_len")
    ws = _ws(workspace_bytes(STAGE_INFLATE, n_sel, IngestConfig()), dev)
    keep = torch.empty(1, dtype=torch.uint8, device=dev) if total == 0 else raw  # a valid pointer even when every chunk is empty
    rc = _lib.hip_lib().hmse_l1_inflate(_ptr(streams), streams.numel(), _ptr(stream_off), _ptr(stream_len), _ptr(kind), _ptr(base),
                                        n_sel, _ptr(raw_off), _ptr(keep), total, _ptr(ok), _ptr(status), ws.data_ptr(), ws.numel(), _stream())
    _check(rc, "hmse_l1_inflate")
    st = int(status.item())
    if check and st:
        raise HmseError(-1, f"hmse_l1_inflate: {int((ok == 0).sum().item())} corrupt record(s), device status {st:#x}")
    return raw, raw_off, ok


def read_assemble(cuts: torch.Tensor, slot_of_chunk: torch.Tensor, raw_off: torch.Tensor, raw: torch.Tensor) -> torch.Tensor:
    """Lay the stored chunks out as the original data: chunk i <- raw bytes of slot_of_chunk[i]. README.md:1635-1669."""
    for t, nm in ((cuts, "cuts"), (slot_of_chunk, "slot_of_chunk"), (raw_off, "raw_off"), (raw, "raw")):
        _require_gpu(t, nm)
    n_chunks = cuts.numel() - 1
    n = int(cuts[-1].item()) if n_chunks > 0 else 0
    out = torch.empty(n, dtype=torch.uint8, device=cuts.device)
    status = torch.zeros(1, dtype=torch.int32, device=cuts.device)
    if n_chunks <= 0 or n == 0:
        return out
    rc = _lib.hip_lib().hmse_read_assemble(_ptr(cuts), n_chunks, _ptr(slot_of_chunk), raw_off.numel() - 1, _ptr(raw_off), _ptr(raw),
                                           _ptr(out), n, _ptr(status), _stream())
    _check(rc, "hmse_read_assemble")
    if int(status.item()):
        raise HmseError(-1, "hmse_read_assemble: chunk map disagrees with the stored lengths")
    return out


def manifest_pack(res, shard: int, n_shards: int, shard_bases, rec_off: torch.Tensor, lba_unit: int, ptr_index: torch.Tensor,
                  blob: torch.Tensor, index: torch.Tensor, chunk_map: torch.Tensor, pointers: torch.Tensor, any_target: bool = False) -> None:
    """hmse_manifest_pack over a ShardResult: blob, ChunkIndex table, chunk map and pointer records written in place.
    README.md:1263-1270, 2182-2189, 1312, 1448."""
    for t, nm in ((res.cuts, "cuts"), (res.uniq_ids, "uniq_ids"), (res.streams, "streams"), (res.stream_off, "stream_off"), (res.kind, "kind"),
                  (rec_off, "rec_off"), (ptr_index, "ptr_index"), (blob, "blob"), (index, "index"), (chunk_map, "chunk_map"), (pointers, "pointers")):
        _require_gpu(t, nm)
    dev = res.cuts.device
    n = res.cuts.numel() - 1
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = _ws(workspace_bytes(STAGE_MANIFEST, n, IngestConfig()), dev)
    base = res.base if res.base is not None else torch.full((res.uniq_ids.numel(),), -1, dtype=torch.int64, device=dev)
    if getattr(res, "base_global", None) is not None:   # dictionaries stored on other shards: -2 (an unresolved DeltaChunk header)
        base = torch.where((res.base_global >= 0) & (base < 0), torch.full_like(base, -2), base)
    sb = None
    if n_shards > 1:
        sb = torch.as_tensor(list(shard_bases), dtype=torch.int64, device=dev)
        if sb.numel() != n_shards:
            raise HmseError(-1, "manifest_pack: one chunk base per shard")
    keep = blob if blob.numel() else torch.empty(1, dtype=torch.uint8, device=dev)
    rc = _lib.hip_lib().hmse_manifest_pack_ex(_ptr(res.streams), _ptr(res.stream_off), _ptr(res.kind), _ptr(base), _ptr(res.uniq_ids),
                                          res.uniq_ids.numel(), _ptr(res.digests), _ptr(res.refcount), _ptr(res.cuts), n, _ptr(res.first_occ),
                                          int(res.chunk_base), shard, _ptr(sb), n_shards, 1 if any_target else 0, _ptr(rec_off), lba_unit, _ptr(ptr_index), _ptr(keep),
                                          blob.numel(), _ptr(index), _ptr(chunk_map), _ptr(pointers) if pointers.numel() else None,
                                          pointers.shape[0], _ptr(status), ws.data_ptr(), ws.numel(), _stream())
    _check(rc, "hmse_manifest_pack")
    st = int(status.item())
    if st:
        raise HmseError(-2, f"hmse_manifest_pack device status {st:#x}")


def stream_batch_workspace_bytes(batch_bytes: int, cfg: IngestConfig) -> int:
    c = cfg.to_c()
    return int(_lib.hip_lib().hmse_stream_batch_workspace_bytes(int(batch_bytes), C.byref(c)))


def stream_workspace(batch_bytes: int, cfg: IngestConfig, device) -> torch.Tensor:
    """A workspace for hmse_stream_batch / hmse_stream_piece_*: allocated and prepared once (hmse_stream_workspace_init: the MinHash
    memo table inside persists across the stream's batches), then passed to every batch."""
    ws = torch.empty(stream_batch_workspace_bytes(batch_bytes, cfg), dtype=torch.uint8, device=device)
    c = cfg.to_c()
    _check(_lib.hip_lib().hmse_stream_workspace_init(ws.data_ptr(), ws.numel(), int(batch_bytes), C.byref(c), _stream()), "hmse_stream_workspace_init")
    return ws


def stream_batch(data: torch.Tensor, batch_bytes: int, seg_off: torch.Tensor, cfg: IngestConfig, state: torch.Tensor, cuts_all: torch.Tensor,
                 max_chunks: int, digests_all: torch.Tensor, first_occ: torch.Tensor, refcount: torch.Tensor, l3_table: torch.Tensor,
                 uniq_all: torch.Tensor, max_unique: int, sig_all: torch.Tensor, band_keys: torch.Tensor, base_all: torch.Tensor,
                 lsh_tables: torch.Tensor, kind_all: torch.Tensor, stream_off_all: torch.Tensor, out: torch.Tensor, ws: torch.Tensor,
                 data_origin: int = 0) -> None:
    """hmse_stream_batch: the whole per-batch chain (L2 -> L3 -> index -> L4 -> band tables -> L1) enqueued without a host
    read; capturable into a hipGraph (hmse_amd/stream.py).  README.md:1519-1580."""
    for t, nm in ((data, "data"), (seg_off, "seg_off"), (state, "state"), (cuts_all, "cuts"), (digests_all, "digests"), (first_occ, "first_occ"),
                  (refcount, "refcount"), (l3_table, "l3_table"), (uniq_all, "uniq"), (sig_all, "sig"), (band_keys, "band_keys"), (base_all, "base"),
                  (lsh_tables, "lsh_tables"), (kind_all, "kind"), (stream_off_all, "stream_off"), (out, "out"), (ws, "ws")):
        _require_gpu(t, nm)
    c = cfg.to_c()
    # data_origin: `data` holds the stream's bytes [data_origin, data_origin + data.numel()) — the resident WINDOW of a stream longer than
    # the buffer (stream.StreamIngest(window_bytes=)); the chain addresses bytes by stream offset, so it gets the buffer's address minus the origin
    rc = _lib.hip_lib().hmse_stream_batch(data.data_ptr() - int(data_origin), int(data_origin) + data.numel(), int(batch_bytes), _ptr(seg_off), seg_off.numel() - 1, C.byref(c), _ptr(state),
                                         _ptr(cuts_all), int(max_chunks), _ptr(digests_all), _ptr(first_occ), _ptr(refcount), _ptr(l3_table),
                                         l3_table.numel(), _ptr(uniq_all), int(max_unique), _ptr(sig_all), _ptr(band_keys), _ptr(base_all),
                                         _ptr(lsh_tables), lsh_tables.shape[1], _ptr(kind_all), _ptr(stream_off_all), _ptr(out), out.numel(),
                                         ws.data_ptr(), ws.numel(), _stream())
    _check(rc, "hmse_stream_batch")


def stream_row_bytes(cap_bytes: int, cfg: IngestConfig) -> int:
    c = cfg.to_c()
    return int(_lib.hip_lib().hmse_stream_row_bytes(int(cap_bytes), C.byref(c)))


def stream_piece_hash(data: torch.Tensor, piece_bytes: int, cap_bytes: int, seg_off: torch.Tensor | None, cfg: IngestConfig, state: torch.Tensor,
                      cuts_all: torch.Tensor, max_chunks: int, row: torch.Tensor, ws: torch.Tensor) -> None:
    """hmse_stream_piece_hash: phase A of a multi-rank stream's batch (L2 + L3 hash of this rank's piece -> its exchange row).
    README.md:1519-1540."""
    for t, nm in ((data, "data"), (state, "state"), (cuts_all, "cuts"), (row, "row"), (ws, "ws")):
        _require_gpu(t, nm)
    if seg_off is not None:
        _require_gpu(seg_off, "seg_off")
    c = cfg.to_c()
    rc = _lib.hip_lib().hmse_stream_piece_hash(_ptr(data), data.numel(), int(piece_bytes), int(cap_bytes), _ptr(seg_off),
                                              0 if seg_off is None else seg_off.numel() - 1, C.byref(c), _ptr(state), _ptr(cuts_all), int(max_chunks),
                                              _ptr(row), ws.data_ptr(), ws.numel(), _stream())
    _check(rc, "hmse_stream_piece_hash")


def stream_piece_encode(data: torch.Tensor, piece_bytes: int, cap_bytes: int, cfg: IngestConfig, state: torch.Tensor, rows: torch.Tensor, world: int,
                        rank: int, cuts_all: torch.Tensor, gidx: torch.Tensor | None, digests_g: torch.Tensor, max_chunks_g: int,
                        first_occ_g: torch.Tensor, refcount_g: torch.Tensor, l3_table: torch.Tensor, uniq_all: torch.Tensor, max_unique: int,
                        sig_all: torch.Tensor, band_keys: torch.Tensor, base_all: torch.Tensor, lsh_tables: torch.Tensor, kind_all: torch.Tensor,
                        stream_off_all: torch.Tensor, out: torch.Tensor, ws: torch.Tensor) -> None:
    """hmse_stream_piece_encode: phase B (gathered rows -> global index -> this rank's stored chunks -> L4 -> L1).  README.md:1538-1580."""
    for t, nm in ((data, "data"), (state, "state"), (rows, "rows"), (cuts_all, "cuts"), (digests_g, "digests"), (first_occ_g, "first_occ"),
                  (refcount_g, "refcount"), (l3_table, "l3_table"), (uniq_all, "uniq"), (sig_all, "sig"), (band_keys, "band_keys"), (base_all, "base"),
                  (lsh_tables, "lsh_tables"), (kind_all, "kind"), (stream_off_all, "stream_off"), (out, "out"), (ws, "ws")):
        _require_gpu(t, nm)
    if gidx is not None:
        _require_gpu(gidx, "gidx")
    if rows.numel() != world * stream_row_bytes(cap_bytes, cfg):
        raise HmseError(-1, "stream_piece_encode: rows must hold one exchange row per rank")
    c = cfg.to_c()
    rc = _lib.hip_lib().hmse_stream_piece_encode(_ptr(data), data.numel(), int(piece_bytes), int(cap_bytes), C.byref(c), _ptr(state), _ptr(rows), int(world),
                                                int(rank), _ptr(cuts_all), _ptr(gidx), _ptr(digests_g), int(max_chunks_g), _ptr(first_occ_g),
                                                _ptr(refcount_g), _ptr(l3_table), l3_table.numel(), _ptr(uniq_all), int(max_unique), _ptr(sig_all),
                                                _ptr(band_keys), _ptr(base_all), _ptr(lsh_tables), lsh_tables.shape[1], _ptr(kind_all),
                                                _ptr(stream_off_all), _ptr(out), out.numel(), ws.data_ptr(), ws.numel(), _stream())
    _check(rc, "hmse_stream_piece_encode")


# ---- global L4 of a multi-rank stream as captured phases (include/hmse.h: hmse_gl4) ---------------------------------------------
def stream_sig_cap(cap_bytes: int, cfg: IngestConfig) -> int:
    c = cfg.to_c()
    return int(_lib.hip_lib().hmse_stream_sig_cap(int(cap_bytes), C.byref(c)))


def stream_sig_row_bytes(cap_bytes: int, cfg: IngestConfig) -> int:
    c = cfg.to_c()
    return int(_lib.hip_lib().hmse_stream_sig_row_bytes(int(cap_bytes), C.byref(c)))


def stream_piece_sign(data, piece_bytes, cap_bytes, cfg, state, rows, world, rank, cuts_all, gidx, digests_g, max_chunks_g, first_occ_g, refcount_g,
                      l3_table, uniq_all, max_unique, sig_all, sig_row, ws) -> None:
    """hmse_stream_piece_sign: gathered digest rows -> global index -> this rank's new stored chunks -> MinHash -> its signature row."""
    for t, nm in ((data, "data"), (state, "state"), (rows, "rows"), (cuts_all, "cuts"), (digests_g, "digests"), (first_occ_g, "first_occ"),
                  (refcount_g, "refcount"), (l3_table, "l3_table"), (uniq_all, "uniq"), (sig_all, "sig"), (sig_row, "sig_row"), (ws, "ws")):
        _require_gpu(t, nm)
    c = cfg.to_c()
    rc = _lib.hip_lib().hmse_stream_piece_sign(_ptr(data), data.numel(), int(piece_bytes), int(cap_bytes), C.byref(c), _ptr(state), _ptr(rows), int(world),
                                              int(rank), _ptr(cuts_all), _ptr(gidx), _ptr(digests_g), int(max_chunks_g), _ptr(first_occ_g), _ptr(refcount_g),
                                              _ptr(l3_table), l3_table.numel(), _ptr(uniq_all), int(max_unique), _ptr(sig_all), _ptr(sig_row),
                                              ws.data_ptr(), ws.numel(), _stream())
    _check(rc, "hmse_stream_piece_sign")


def stream_piece_bases(cap_bytes, cfg, state, sig_rows, gl4, uniq_all, band_keys, base_all, ws) -> None:
    """hmse_stream_piece_bases: gathered signature rows -> global band tables -> dictionaries of this rank's new stored chunks + requests.
    `gl4`: a _lib.HmseGl4 (kept alive by the caller together with the tensors it points to)."""
    for t, nm in ((state, "state"), (sig_rows, "sig_rows"), (uniq_all, "uniq"), (band_keys, "band_keys"), (base_all, "base"), (ws, "ws")):
        _require_gpu(t, nm)
    c = cfg.to_c()
    rc = _lib.hip_lib().hmse_stream_piece_bases(int(cap_bytes), C.byref(c), _ptr(state), _ptr(sig_rows), C.byref(gl4), _ptr(uniq_all), _ptr(band_keys),
                                               _ptr(base_all), ws.data_ptr(), ws.numel(), _stream())
    _check(rc, "hmse_stream_piece_bases")


def stream_piece_encode_g(data, piece_bytes, cap_bytes, cfg, state, gstate, cuts_all, kind_all, stream_off_all, out, ws) -> None:
    """hmse_stream_piece_encode_g: DEFLATE of the new stored chunks with the dictionaries resolved by stream_piece_bases, tails, states advanced."""
    for t, nm in ((data, "data"), (state, "state"), (gstate, "gstate"), (cuts_all, "cuts"), (kind_all, "kind"), (stream_off_all, "stream_off"), (out, "out"), (ws, "ws")):
        _require_gpu(t, nm)
    c = cfg.to_c()
    rc = _lib.hip_lib().hmse_stream_piece_encode_g(_ptr(data), data.numel(), int(piece_bytes), int(cap_bytes), C.byref(c), _ptr(state), _ptr(gstate),
                                                  _ptr(cuts_all), _ptr(kind_all), _ptr(stream_off_all), _ptr(out), out.numel(), ws.data_ptr(), ws.numel(), _stream())
    _check(rc, "hmse_stream_piece_encode_g")
