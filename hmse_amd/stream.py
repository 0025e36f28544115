"""Streaming front end (BASELINE.json configs[4]): the corpus arrives in batches, host -> HBM copies of
batch k+1 overlap the kernels of batch k, and the index persists across batches.

Every batch runs the same C-ABI stages as ingest_shard; what persists between batches lives in HBM:
  * the raw bytes of everything ingested so far (288 GB of HBM hold the whole 4 x 10 GB configuration), so a chunk of
    an earlier batch can serve as DEFLATE dictionary for a later one (hmse_l1_deflate_ex, HMSE_DEFLATE_BASE_IS_CHUNK_ID);
  * all cut points, digests and the L3 hash table over them (README.md:1263-1270, 1288-1292 — first occurrences are GLOBAL
    over the stream), signatures of the stored chunks and the four L4 band tables (a base is the earliest stored chunk that
    shares a band, whichever batch it came from).  Tables and arrays are allocated once and updated IN PLACE: a batch
    inserts and looks up only its own chunks, so its cost does not grow with the history.
With batches that are whole multiples of the segment size, every output equals what ingest_shard returns for the
concatenated input, bit for bit (tests/test_gpu_stream.py) — batching is a schedule, not a different result.
"""
from __future__ import annotations

import torch

from . import ops
from .config import LAYER_L1, LAYER_L2, LAYER_L3, LAYER_L4, IngestConfig
from .ingest import ShardResult, shard_stats


class StreamIngest:
    def __init__(self, cfg: IngestConfig, capacity_bytes: int, device, max_chunks: int | None = None, graph: bool = False,
                 stream_capacity: int | None = None, poll_status_every: int = 0, window_bytes: int | None = None):
        """graph=True: every batch runs as ONE enqueue of the device-count chain (hmse_stream_batch: no host read between
        stages); from the second batch of a given size on, that enqueue is a replay of a hipGraph captured once for the size.
        Results are identical to graph=False.  stream_capacity: bytes reserved for the DEFLATE streams in that mode
        (default: half the corpus capacity + 64 MiB; overflow is reported, never written).  poll_status_every = N > 0 (graph
        mode): after every N-th batch the chain's sticky status word is copied to pinned host memory WITHOUT waiting for it; a
        later push() that finds the copy complete and non-zero raises then instead of at finish() — a failed batch and all
        later ones are no-ops on the device either way, the poll only tells a long stream early."""
        if cfg.layers != (LAYER_L1 | LAYER_L2 | LAYER_L3 | LAYER_L4):
            raise ValueError("StreamIngest runs the full L1-L4 pipeline")
        self.cfg, self.dev = cfg, device
        # window_bytes (round 4; the reference's loop runs "until input exhausted" in fixed memory, README.md:262-276, 1579-1580): only the last
        # `window_bytes` of the stream stay resident.  The stream is cut into WINDOWS — runs of whole batches of at most window_bytes —; the raw
        # bytes of a window replace those of the one before, and a dictionary (L4 base) is only ever a chunk of the SAME window: the band
        # tables are cleared at a window's start.  Exact dedupe (L3: digests only) spans the whole stream as before.  capacity_bytes then
        # only sizes the index arrays.  The CPU oracle with the same windows gives the same records (tests/test_gpu_stream.py).
        self.window = int(window_bytes) if window_bytes else 0
        if self.window:
            if not graph:
                raise ValueError("window_bytes needs the device-count chain (graph=True)")
            if self.window % cfg.seg_size:
                raise ValueError("window_bytes must be a multiple of seg_size")
        self._wcopy0 = 0            # stream offset of the buffer's first byte: the window batches are being COPIED into ...
        self._wproc0 = 0            # ... and the one the chain is working on
        self._wstarts: list[int] = []   # stream offsets at which a new window starts (copy side -> processing side)
        self.window_starts: list[int] = [0]
        self.data = torch.empty(int(self.window or capacity_bytes), dtype=torch.uint8, device=device)
        self.n_bytes = 0            # bytes whose host -> HBM copy has been issued
        self.n_done = 0             # bytes processed
        self.copy_stream = torch.cuda.Stream(device=device)
        self.pending: list[tuple[int, int, torch.cuda.Event]] = []   # (offset, length, copy-done event)
        # The index persists in HBM across batches (README.md:1288-1292, 1554-1576): every per-chunk array is allocated
        # once for `max_chunks` (default: twice the expected count at the configured average chunk size) and only its tail
        # is written by a batch; the L3 table and the four L4 band tables are updated in place (hmse_l3_index_update,
        # hmse_l4_lsh_update), so a batch costs the same whatever came before it.
        self.max_chunks = int(max_chunks or (capacity_bytes // max(1, cfg.avg_size // 2) + capacity_bytes // cfg.seg_size + 64))
        self.max_unique = min(self.max_chunks, 1 << 23)              # the band tables hold 2^24 slots at load <= 0.5
        mc, mu = self.max_chunks, self.max_unique
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=device)
        self._cuts = z(mc + 1, torch.int64)
        self._digests = torch.empty((mc, 32), dtype=torch.uint8, device=device)
        self._first_occ = z(mc, torch.int64)
        self._refcount = z(mc, torch.int32)
        self._uniq = z(mu, torch.int64)         # global chunk index of every stored chunk, ascending
        self._sig = torch.empty((mu, cfg.n_hashes), dtype=torch.int32, device=device)
        self._band_keys = z((mu, cfg.bands), torch.int32)
        self._base = z(mu, torch.int64)         # per stored chunk: slot of its dictionary chunk or -1
        self._kind = z(mu, torch.uint8)
        self._stream_off = z(mu + 1, torch.int64)
        self._l3_table = torch.empty(ops.l3_index_slots(mc), dtype=torch.int32, device=device)
        self._lsh_tables = torch.empty((cfg.bands, ops.l4_lsh_slots(mu)), dtype=torch.int32, device=device)
        self.n_chunks = 0
        self.n_unique = 0
        self.stream_bytes = 0
        self.poll_status_every = int(poll_status_every)
        self._polls: list[tuple[torch.Tensor, torch.cuda.Event, int]] = []   # (pinned status word, copy-done event, batch number)
        self._n_chain_batches = 0
        self.stream_parts: list[torch.Tensor] = []
        self._ws = None             # DEFLATE workspace, kept across batches
        self.graph = bool(graph)
        if self.graph:
            self._state = torch.zeros(16, dtype=torch.int64, device=device)
            self._streams = torch.empty(int(stream_capacity or (capacity_bytes // 2 + (64 << 20))), dtype=torch.uint8, device=device)
            self._graphs: dict[int, tuple] = {}     # batch bytes -> (CUDAGraph, workspace, seg_off) once captured; None after the eager first batch
            self._state_dirty = True                # host counters -> device state before the next chain call

    # views of the filled part of the index
    cuts = property(lambda self: self._cuts[: self.n_chunks + 1])
    digests = property(lambda self: self._digests[: self.n_chunks])
    first_occ = property(lambda self: self._first_occ[: self.n_chunks])
    refcount = property(lambda self: self._refcount[: self.n_chunks])
    uniq_ids = property(lambda self: self._uniq[: self.n_unique])
    sig = property(lambda self: self._sig[: self.n_unique])
    band_keys = property(lambda self: self._band_keys[: self.n_unique])
    base = property(lambda self: self._base[: self.n_unique])
    kind = property(lambda self: self._kind[: self.n_unique])
    stream_off = property(lambda self: self._stream_off[: self.n_unique + 1])

    def index_sidecar(self) -> bytes:
        """The L4 index for a later resume(): the reference's band tables (per band {band_hash u16, count u16} + 3-byte ids,
        README.md:1937-1945) followed by the signatures, which the content-addressed tables in HBM are rebuilt from."""
        from . import bandtable
        return bandtable.write_band_tables(self.band_keys.cpu().numpy(), self.cfg.band_bits, signatures=self.sig.cpu().numpy())

    @staticmethod
    def resume(m, cfg: IngestConfig, capacity_bytes: int, device, band_tables: bytes | None = None, verify: bool = True,
               max_chunks: int | None = None, graph: bool = False) -> "StreamIngest":
        """Incremental ingest against an existing store.  The raw bytes come back through the GPU read path
        (any stored chunk may become a dictionary); the INDEX is loaded, not recomputed: digests from the ChunkIndex records
        go straight into the L3 table, band keys and signatures from the band-table sidecar (index_sidecar()) go straight
        into the L4 tables.  Without a sidecar the signatures are recomputed from the restored bytes.  `verify` re-checks
        every chunk's SHA-256 while reading (the reference's read-side gate, README.md:1329).  Batches pushed afterwards
        dedupe and delta against the restored chunks exactly as if the whole history had been one stream."""
        import numpy as np
        from . import bandtable, read
        from .config import KIND_POINTER
        st = StreamIngest(cfg, capacity_bytes, device, max_chunks, graph=graph)
        data = read.read_manifest(m, device, verify=verify)
        n = data.numel()
        if n % cfg.seg_size:
            raise ValueError("the stored stream ends inside a segment: nothing can be appended to it")
        if n > st.data.numel():
            raise ValueError("stream capacity exceeded")
        st.data[:n] = data
        p = read.parse_manifest(m)
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).copy()).to(dt).to(device)
        nc, nu = len(m.chunk_map), len(m.index)
        if nc > st.max_chunks or nu > st.max_unique:
            raise ValueError("stream index capacity exceeded")
        torch.cumsum(t(p["raw_len"][m.chunk_map["slot"]], torch.int64), 0, out=st._cuts[1: nc + 1])
        st._digests[:nc] = t(m.index["sha256"][m.chunk_map["slot"]], torch.uint8)
        ops.l3_index_update(st._digests, 0, nc, st._first_occ, st._refcount, st._l3_table)
        own = np.nonzero(m.chunk_map["kind"] != KIND_POINTER)[0]
        if len(own) != nu or not np.array_equal(m.chunk_map["slot"][own], np.arange(nu)):
            raise ValueError("manifest index and chunk map disagree on the stored chunks")
        st._uniq[:nu] = t(own, torch.int64)
        st.n_chunks, st.n_unique = nc, nu
        idx = torch.arange(nc, dtype=torch.int64, device=device)
        if not torch.equal(idx[st.first_occ == idx], st.uniq_ids):
            raise ValueError("the stored chunks are not the first occurrences of their digests")
        if band_tables is not None:
            keys, sig = bandtable.read_signatures(band_tables)
            if sig is None or sig.shape != (nu, cfg.n_hashes) or keys.shape != (nu, cfg.bands):
                raise ValueError("the band-table sidecar does not belong to this manifest / configuration")
            st._sig[:nu] = t(sig.view(np.int32), torch.int32)
            st._band_keys[:nu] = t(keys.view(np.int32), torch.int32)
            ops.l4_lsh_update(st._sig, 0, nu, cfg, st._band_keys, st._base, st._lsh_tables, keys_given=True)
        else:
            st._sig[:nu] = ops.l4_minhash(st.data[:n], st.cuts, cfg, st.uniq_ids)
            ops.l4_lsh_update(st._sig, 0, nu, cfg, st._band_keys, st._base, st._lsh_tables)
        st._kind[:nu] = t(p["kind"], torch.uint8)
        # dense copy of the stored streams (the blob aligns records to lba_unit and prefixes DELTA records with a header)
        s_len = t(p["stream_len"], torch.int64)
        torch.cumsum(s_len, 0, out=st._stream_off[1: nu + 1])
        st.stream_bytes = int(st._stream_off[nu].item())
        src = torch.arange(st.stream_bytes, dtype=torch.int64, device=device) + \
            torch.repeat_interleave(t(p["stream_off"], torch.int64) - st._stream_off[:nu], s_len)
        st.stream_parts = [t(m.blob, torch.uint8)[src]]
        # the records must agree with what this build derives from the loaded index
        mb = t(p["base"], torch.int64)
        if not bool(((st.kind != 2) | (mb == st.base)).all()):
            raise ValueError("a DELTA record's dictionary is not the LSH base of its chunk under this configuration")
        st.n_bytes = st.n_done = n
        return st

    # ------------------------------------------------------------------ feeding
    def push(self, host_batch: torch.Tensor) -> None:
        """Issue the host -> HBM copy of the next batch (asynchronous when the tensor is pinned), then process every
        batch whose copy was issued earlier: the copy of this batch overlaps the kernels of the previous one."""
        n = host_batch.numel()
        self._check_polls()
        if self.n_bytes % self.cfg.seg_size:
            raise ValueError("only the last batch may be a partial segment")
        if self.window:
            if n > self.window:
                raise ValueError("a batch may not exceed window_bytes")
            if self.n_bytes + n > self._wcopy0 + self.window:
                # this batch opens a new window at the buffer's start: everything enqueued so far may still read the old window's bytes
                # (dictionaries), so it is processed first and the copy waits for it — the one batch per window whose copy does not overlap
                while self.pending:
                    self._process(*self.pending.pop(0))
                done = torch.cuda.Event()
                done.record()
                self.copy_stream.wait_event(done)
                self._wcopy0 = self.n_bytes
                self._wstarts.append(self.n_bytes)
        elif self.n_bytes + n > self.data.numel():
            raise ValueError("stream capacity exceeded")
        ev = torch.cuda.Event()
        with torch.cuda.stream(self.copy_stream):
            self.data[self.n_bytes - self._wcopy0: self.n_bytes - self._wcopy0 + n].copy_(host_batch, non_blocking=True)
            ev.record(self.copy_stream)
        while self.pending:  # batches issued before this one
            self._process(*self.pending.pop(0))
        self.pending.append((self.n_bytes, n, ev))
        self.n_bytes += n

    def finish(self) -> ShardResult:
        while self.pending:
            self._process(*self.pending.pop(0))
        if self.graph and not self._state_dirty:
            self._read_state()
            streams = self._streams[: self.stream_bytes]
            if self.stream_parts:   # streams restored by resume() precede the chain's
                streams = torch.cat(self.stream_parts + [self._streams[self._chain_s0: self.stream_bytes]])
        else:
            streams = torch.cat(self.stream_parts) if self.stream_parts else torch.empty(0, dtype=torch.uint8, device=self.dev)
        res = ShardResult(self.n_done, self.cuts, self.digests, 0, self.n_chunks, self.first_occ, self.refcount, self.uniq_ids,
                          self.sig, self.band_keys, self.base, streams, self.stream_off, self.kind)
        res.stats = shard_stats(res)
        return res

    # ------------------------------------------------------------------ one batch, device-count chain (graph=True)
    def _read_state(self) -> None:
        """The ONE host read of the chain: counts after the batches enqueued so far."""
        st = self._state.tolist()
        if st[7]:
            raise ValueError(f"streaming chain status {st[7]:#x}: bit0 chunk capacity, bit1 stored-chunk capacity, bit2 L2, bit3 exchange row, bit4 workspace not initialised, bit5 state block inconsistent ([8] != [1] on one rank), "
                             "bits 8.. DEFLATE (0x100 stream capacity, 0x200 workspace); the failing batch and every later one were dropped "
                             f"({st[0]} bytes / {st[1]} chunks are intact)")
        self.n_done, self.n_chunks, self.n_unique, self.stream_bytes = st[0], st[1], st[3], st[5]

    def _chain_call(self, n: int, seg_off: torch.Tensor, ws: torch.Tensor) -> None:
        ops.stream_batch(self.data, n, seg_off, self.cfg, self._state, self._cuts, self.max_chunks, self._digests, self._first_occ, self._refcount,
                         self._l3_table, self._uniq, self.max_unique, self._sig, self._band_keys, self._base, self._lsh_tables, self._kind,
                         self._stream_off, self._streams, ws, data_origin=self._wproc0)

    def _process_chain(self, off: int, n: int, copied: torch.cuda.Event) -> None:
        dev = self.dev
        if self._state_dirty:   # first chain call (possibly after resume()): host counters -> device state
            self._chain_s0 = self.stream_bytes
            if self.n_chunks == 0:
                ops.l3_index_update(self._digests, 0, 0, self._first_occ, self._refcount, self._l3_table)      # clears the table
                ops.l4_lsh_update(self._sig, 0, 0, self.cfg, self._band_keys, self._base, self._lsh_tables)     # clears the tables
            self._state.copy_(torch.tensor([off, self.n_chunks, 0, self.n_unique, 0, self.stream_bytes, 0, 0, self.n_chunks] + [0] * 7, dtype=torch.int64))
            self._state_dirty = False
        torch.cuda.current_stream().wait_event(copied)
        if self._wstarts and off == self._wstarts[0]:
            # a new window: dictionaries come from this window only (band tables cleared; signatures and band keys of earlier windows
            # stay in their arrays, nothing points at them any more), and the captured graphs hold the old window's data address
            self._wproc0 = self._wstarts.pop(0)
            self.window_starts.append(self._wproc0)
            ops.l4_lsh_update(self._sig, 0, 0, self.cfg, self._band_keys, self._base, self._lsh_tables)
            self._graphs = {k: (None, v[1], v[2]) for k, v in self._graphs.items()}
        entry = self._graphs.get(n, 0)
        if entry == 0:          # first batch of this size: plain enqueue (also sets the kernels' attributes before any capture)
            seg_off = ops.segment_offsets(n, self.cfg.seg_size, dev)
            ws = ops.stream_workspace(n, self.cfg, dev)
            self._chain_call(n, seg_off, ws)
            self._graphs[n] = (None, ws, seg_off)
        else:
            g, ws, seg_off = entry
            if g is None:       # second batch of this size: capture the enqueue once ...
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._chain_call(n, seg_off, ws)
                self._graphs[n] = (g, ws, seg_off)
            g.replay()          # ... and from then on every batch of this size is one graph launch
        self.n_done = off + n
        self._n_chain_batches += 1
        if self.poll_status_every and self._n_chain_batches % self.poll_status_every == 0:
            host = torch.zeros(1, dtype=torch.int64).pin_memory()
            host.copy_(self._state[7:8], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._polls.append((host, ev, self._n_chain_batches))

    def _check_polls(self) -> None:
        """Completed status polls (never waits): raise on the first non-zero one."""
        while self._polls and self._polls[0][1].query():
            host, _, nb = self._polls.pop(0)
            st = int(host.item())
            if st:
                raise ValueError(f"streaming chain status {st:#x} (polled after batch {nb}): bit0 chunk capacity, bit1 stored-chunk capacity, "
                                 "bit2 L2, bits 8.. DEFLATE (0x100 stream capacity, 0x200 workspace); the failing batch and every later one were dropped")

    # ------------------------------------------------------------------ one batch, host-sized launches (graph=False)
    def _process(self, off: int, n: int, copied: torch.cuda.Event) -> None:
        if self.graph:
            return self._process_chain(off, n, copied)
        cfg, dev = self.cfg, self.dev
        torch.cuda.current_stream().wait_event(copied)
        batch = self.data[off: off + n]
        all_data = self.data[: off + n]
        # L2 on the batch (segments restart at batch-local multiples of seg_size == global ones), offsets made global
        cuts_b = ops.l2_cdc(batch, cfg)
        n_old, n_new = self.n_chunks, cuts_b.numel() - 1
        if n_old + n_new > self.max_chunks:
            raise ValueError("stream index capacity exceeded (max_chunks)")
        self._cuts[n_old + 1: n_old + n_new + 1] = cuts_b[1:] + off
        cuts_all = self._cuts[: n_old + n_new + 1]
        # L3: digests of the new chunks join the persistent table; first occurrences are global over the stream
        self._digests[n_old: n_old + n_new] = ops.l3_sha256(all_data, self._cuts[n_old: n_old + n_new + 1])
        ops.l3_index_update(self._digests, n_old, n_new, self._first_occ, self._refcount, self._l3_table)
        idx_new = torch.arange(n_old, n_old + n_new, dtype=torch.int64, device=dev)
        uniq_new = idx_new[self._first_occ[n_old: n_old + n_new] == idx_new]
        u_old, nu = self.n_unique, int(uniq_new.numel())
        if u_old + nu > self.max_unique:
            raise ValueError("stream index capacity exceeded (stored chunks)")
        self._uniq[u_old: u_old + nu] = uniq_new
        # L4: signatures of the new stored chunks join the persistent band tables; a base is the earliest stored chunk
        # sharing a band, whichever batch it came from
        self._sig[u_old: u_old + nu] = ops.l4_minhash(all_data, cuts_all, cfg, uniq_new)
        ops.l4_lsh_update(self._sig, u_old, nu, cfg, self._band_keys, self._base, self._lsh_tables)
        base_new = self._base[u_old: u_old + nu]
        # L1: the dictionary may be a chunk of an earlier batch -> bases as chunk indices
        base_chunk = torch.where(base_new >= 0, self._uniq[base_new.clamp(min=0)], base_new)
        # one workspace for every batch, sized for the worst case of this batch (every chunk stored, every one with a dictionary):
        # a fresh multi-GB allocation per batch synchronises the device and stalls the copy/compute overlap
        worst = (5 * n + 1600 * n_new) + ops.workspace_bytes(ops.STAGE_DEFLATE, n_new, cfg) + (1 << 20)
        if self._ws is None or self._ws.numel() < worst:
            self._ws = None
            self._ws = torch.empty(worst, dtype=torch.uint8, device=dev)
        streams_b, off_b, kind_b = ops.l1_deflate(all_data, cuts_all, cfg, uniq_new, base_chunk, base_is_chunk_id=True, ws=self._ws)
        self.stream_parts.append(streams_b)
        self._stream_off[u_old + 1: u_old + nu + 1] = off_b[1:] + self.stream_bytes
        self.stream_bytes += int(streams_b.numel())
        self._kind[u_old: u_old + nu] = kind_b
        self.n_chunks, self.n_unique = n_old + n_new, u_old + nu
        self.n_done = off + n
