"""Global L4 for a multi-rank stream as hipGraph-captured phases (round 4; SURVEY.md §8f-3 "cross-GPU base-chunk fetch over xGMI P2P for
global L4 (config 5)"; BASELINE.json configs[4] "hipGraph-captured per-batch pipeline"; the reference has ONE set of band tables,
README.md:1375-1383, and one batch loop, README.md:1519-1580).

`stream_dist.GlobalL4StreamIngest` (round 3) gives an N-rank stream the records of the one-rank run, but enqueues every stage eagerly and
sizes three ragged exchanges per batch on the host (2.1-2.3 x the shard-local chain in rehearsal).  Here everything that can be
fixed-size is: per batch and rank

    phase A   hmse_stream_piece_hash      L2 + L3 hash of the piece -> digest row          | all-gather (fixed-size rows)
    phase B1  hmse_stream_piece_sign      global index, new stored chunks, MinHash -> signature row | all-gather (fixed-size rows)
    phase B2  hmse_stream_piece_bases     global band tables -> dictionaries (own chunk ids / remote requests)
    --        remote dictionaries fetched (ingest.fetch_chunks_routed: the ONE step left eager; skipped with one rank)
    phase B3  hmse_stream_piece_encode_g  DEFLATE, tails, states advanced

each phase captured into a hipGraph at its second use and replayed afterwards; the host reads ONE small array per batch (the request counts,
only when world > 1).  The result of every rank equals `GlobalL4StreamIngest`'s (tests/test_gpu_stream_gl4.py: 2 and 3 ranks in lock step
against the oracle's single pass over the logical stream).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib, ops
from .config import LAYER_L1, LAYER_L2, LAYER_L3, LAYER_L4, IngestConfig
from .ingest import ShardResult, shard_stats
from .stream_dist import all_gather_rows, deal_batch


class GraphGlobalL4StreamIngest:
    """This rank's side of a global-L4 stream over `world` ranks, every phase a hipGraph replay.  COLLECTIVE: every rank pushes its
    piece of every global batch, in the same order (an empty tensor if it has no bytes in a batch)."""

    def __init__(self, cfg: IngestConfig, capacity_bytes: int, piece_bytes: int, device, world: int, rank: int, group=None,
                 max_chunks: int | None = None, max_chunks_global: int | None = None, stream_capacity: int | None = None,
                 ghost_bytes: int | None = None, graph: bool = True, always_exchange: bool = False):
        if cfg.layers != (LAYER_L1 | LAYER_L2 | LAYER_L3 | LAYER_L4):
            raise ValueError("GraphGlobalL4StreamIngest runs the full L1-L4 pipeline")
        if piece_bytes <= 0 or piece_bytes % cfg.seg_size:
            raise ValueError("piece_bytes (the nominal piece size) must be a positive multiple of seg_size")
        if not 0 <= rank < world <= 256:
            raise ValueError("0 <= rank < world <= 256")
        self.cfg, self.dev, self.world, self.rank, self.group = cfg, device, int(world), int(rank), group
        self.cap_bytes = int(piece_bytes)
        self.capacity = int(capacity_bytes)
        self.ghost_cap = int(ghost_bytes if ghost_bytes is not None else 2 * piece_bytes)
        self.data = torch.empty(self.capacity + self.ghost_cap, dtype=torch.uint8, device=device)   # [0, capacity): this rank's pieces; behind: fetched dictionaries
        self.n_bytes = 0
        self.copy_stream = torch.cuda.Stream(device=device)
        self.pending: list = []
        per = max(1, cfg.avg_size // 2)
        self.max_chunks = int(max_chunks or (capacity_bytes // per + capacity_bytes // cfg.seg_size + 64))
        self.max_chunks_g = int(max_chunks_global or self.max_chunks * world)
        self.max_unique = min(self.max_chunks, 1 << 23)
        self.max_stored_g = min(self.max_chunks_g, 1 << 23)
        self.sig_cap = ops.stream_sig_cap(self.cap_bytes, cfg)
        mc, mg, mu, mug = self.max_chunks, self.max_chunks_g, self.max_unique, self.max_stored_g
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=device)
        # this rank's chunks; the cut array continues behind them with the bounds of the batch's fetched dictionaries ("ghost" chunks)
        self.ghost_chunk0 = mc + 1
        self._cuts = z(mc + 2 + self.sig_cap + 1, torch.int64)
        self._gidx = z(mc, torch.int64)
        self._uniq = z(mu, torch.int64)
        self._sig = torch.empty((mu, cfg.n_hashes), dtype=torch.int32, device=device)
        self._band_keys = z((mu, cfg.bands), torch.int32)
        self._base = z(mu, torch.int64)              # this rank's stored slot of the dictionary, -1: none or remote
        self._base_global = z(mu, torch.int64)       # global stored index of the dictionary
        self._ug = z(mu, torch.int64)                # global stored index of the chunk
        self._kind = z(mu, torch.uint8)
        self._stream_off = z(mu + 1, torch.int64)
        self._streams = torch.empty(int(stream_capacity or (capacity_bytes // 2 + (64 << 20))), dtype=torch.uint8, device=device)
        # the global index and the global band tables: identical on every rank
        self._digests_g = torch.empty((mg, 32), dtype=torch.uint8, device=device)
        self._first_occ_g = z(mg, torch.int64)
        self._refcount_g = z(mg, torch.int32)
        self._l3_table = torch.empty(ops.l3_index_slots(mg), dtype=torch.int32, device=device)
        self._sig_g = torch.empty((mug, cfg.n_hashes), dtype=torch.int32, device=device)
        self._keys_g = z((mug, cfg.bands), torch.int32)
        self._base_g = z(mug, torch.int64)
        self._lsh_tables_g = torch.empty((cfg.bands, ops.l4_lsh_slots(mug)), dtype=torch.int32, device=device)
        self._g_owner = z(mug, torch.int32)
        self._g_local = z(mug, torch.int64)
        ops.l3_index_update(self._digests_g, 0, 0, self._first_occ_g, self._refcount_g, self._l3_table)       # clears the table
        ops.l4_lsh_update(self._sig_g, 0, 0, cfg, self._keys_g, self._base_g, self._lsh_tables_g)               # clears the tables
        self._state = z(16, torch.int64)
        self._gstate = z(16, torch.int64)
        self._ws = ops.stream_workspace(self.cap_bytes, cfg, device)
        self.row_bytes = ops.stream_row_bytes(self.cap_bytes, cfg)
        self.sig_row_bytes = ops.stream_sig_row_bytes(self.cap_bytes, cfg)
        self._row = z(self.row_bytes, torch.uint8)
        self._rows = z(self.world * self.row_bytes, torch.uint8)
        self._sig_row = z(self.sig_row_bytes, torch.uint8)
        self._sig_rows = z(self.world * self.sig_row_bytes, torch.uint8)
        self._req_counts = z(self.world + 1, torch.int64)
        self._req_slots = z(self.sig_cap, torch.int64)
        g = _lib.HmseGl4()
        g.struct_size = C.sizeof(_lib.HmseGl4); g.world = self.world; g.rank = self.rank; g.reserved = 0
        g.sig_cap = self.sig_cap; g.max_stored_g = mug
        g.gstate = self._gstate.data_ptr(); g.sig_g = self._sig_g.data_ptr(); g.band_keys_g = self._keys_g.data_ptr(); g.base_g = self._base_g.data_ptr()
        g.lsh_tables_g = self._lsh_tables_g.data_ptr(); g.lsh_slots_g = self._lsh_tables_g.shape[1]
        g.g_owner = self._g_owner.data_ptr(); g.g_local = self._g_local.data_ptr(); g.ug = self._ug.data_ptr(); g.base_global = self._base_global.data_ptr()
        g.req_counts = self._req_counts.data_ptr(); g.req_slots = self._req_slots.data_ptr(); g.ghost_chunk0 = self.ghost_chunk0
        self._gl4 = g
        self.graph = bool(graph)
        self._graphs: dict[int, list] = {}      # piece bytes -> [seg_off, graph A, B1, B2, B3, uses]
        self._always_exchange = bool(always_exchange)
        self._host_error = None
        self.n_batches = 0
        self.remote_dictionaries = 0
        self.ghost_bytes_fetched = 0

    # ------------------------------------------------------------------ capture / replay
    def _entry(self, n: int) -> list:
        e = self._graphs.get(n)
        if e is None:
            e = [ops.segment_offsets(n, self.cfg.seg_size, self.dev) if n else None, None, None, None, None, 0]
            self._graphs[n] = e
        return e

    def _run(self, which: int, n: int, fn) -> None:
        """First use of a piece size: plain enqueue (sets the kernels' attributes before any capture); second use: capture; then replay."""
        e = self._entry(n)
        if not self.graph or e[5] < 1:
            fn()
            return
        if e[which] is None:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            e[which] = g
        e[which].replay()

    # ------------------------------------------------------------------ the phases of one batch (enqueue only)
    def phase_a(self, n: int) -> torch.Tensor:
        e = self._entry(n)
        self._run(1, n, lambda: ops.stream_piece_hash(self.data, n, self.cap_bytes, e[0], self.cfg, self._state, self._cuts, self.max_chunks, self._row, self._ws))
        return self._row

    def phase_b1(self, n: int, rows: torch.Tensor) -> torch.Tensor:
        if rows.data_ptr() != self._rows.data_ptr() and not (self.world == 1 and rows.data_ptr() == self._row.data_ptr()):
            self._rows.copy_(rows.reshape(-1))
            rows = self._rows
        self._run(2, n, lambda: ops.stream_piece_sign(self.data, n, self.cap_bytes, self.cfg, self._state, rows, self.world, self.rank, self._cuts, self._gidx,
                                                      self._digests_g, self.max_chunks_g, self._first_occ_g, self._refcount_g, self._l3_table, self._uniq,
                                                      self.max_unique, self._sig, self._sig_row, self._ws))
        return self._sig_row

    def phase_b2(self, n: int, sig_rows: torch.Tensor) -> None:
        if sig_rows.data_ptr() != self._sig_rows.data_ptr() and not (self.world == 1 and sig_rows.data_ptr() == self._sig_row.data_ptr()):
            self._sig_rows.copy_(sig_rows.reshape(-1))
            sig_rows = self._sig_rows
        self._run(3, n, lambda: ops.stream_piece_bases(self.cap_bytes, self.cfg, self._state, sig_rows, self._gl4, self._uniq, self._band_keys, self._base, self._ws))

    def phase_b3(self, n: int) -> None:
        self._run(4, n, lambda: ops.stream_piece_encode_g(self.data, n, self.cap_bytes, self.cfg, self._state, self._gstate, self._cuts, self._kind,
                                                          self._stream_off, self._streams, self._ws))
        self._entry(n)[5] += 1
        self.n_batches += 1

    # ------------------------------------------------------------------ the remote dictionaries of a batch (eager)
    def requests(self):
        """(requests per owner rank [world], the owners' stored slots grouped by owner) of the batch phase B2 just resolved — ONE host read."""
        c = self._req_counts.tolist()
        return c[: self.world], self._req_slots[: c[self.world]]

    def serve(self, local_slots: torch.Tensor):
        """Raw bytes of this rank's stored chunks `local_slots` (what a peer's fetch gets): (bytes, lens).  A chunk stored by THIS batch
        can already be a peer's dictionary: phase B1 has appended it."""
        cid = self._uniq[local_slots]
        ln = self._cuts[cid + 1] - self._cuts[cid]
        oc = torch.zeros(cid.numel() + 1, dtype=torch.int64, device=self.dev)
        torch.cumsum(ln, 0, out=oc[1:])
        return (ops.read_assemble(oc, cid, self._cuts, self.data) if cid.numel() else torch.empty(0, dtype=torch.uint8, device=self.dev)), ln

    def write_ghost(self, ghost: torch.Tensor, lens: torch.Tensor) -> None:
        """The fetched dictionaries, in request order, behind this rank's data; their bounds behind its cuts (request j = chunk ghost_chunk0 + j)."""
        gb, k = int(ghost.numel()), int(lens.numel())
        if gb > self.ghost_cap or k > self.sig_cap:
            raise ValueError(f"a batch needs {gb} bytes / {k} remote dictionaries, the ghost area holds {self.ghost_cap} bytes / {self.sig_cap} chunks (ghost_bytes=)")
        if k:
            self.data[self.capacity: self.capacity + gb] = ghost
            g0 = self.ghost_chunk0
            self._cuts[g0] = self.capacity
            self._cuts[g0 + 1: g0 + 1 + k] = self.capacity + torch.cumsum(lens, 0)
            self.remote_dictionaries += k
            self.ghost_bytes_fetched += gb

    # ------------------------------------------------------------------ N processes: the phases joined by collectives
    def _gather(self, row: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        if self.world == 1 and not self._always_exchange:
            return row
        return all_gather_rows(row, self.world, self.group, out=out)

    def push(self, host_piece: torch.Tensor) -> None:
        """COLLECTIVE: issue the host -> HBM copy of this rank's piece of the next global batch, then process the piece pushed before."""
        n = host_piece.numel()
        err = None
        if n > self.cap_bytes:
            err = "a piece may not exceed the stream's nominal piece size"
        elif self.n_bytes % self.cfg.seg_size:
            err = "only a rank's last piece may end inside a segment"
        elif self.n_bytes + n > self.capacity:
            err = "stream capacity exceeded"
        if err:   # refused THROUGH the chain (sticky status bit 6), the rank keeps in step with its peers: see DistStreamIngest.push
            self._host_error = self._host_error or err
            n, host_piece = 0, host_piece[:0]
        ev = torch.cuda.Event()
        with torch.cuda.stream(self.copy_stream):
            if n:
                self.data[self.n_bytes: self.n_bytes + n].copy_(host_piece, non_blocking=True)
            ev.record(self.copy_stream)
        while self.pending:
            self._process(*self.pending.pop(0))
        if err:
            self._state[7:8] |= 64
        self.pending.append((self.n_bytes, n, ev))
        self.n_bytes += n

    def _process(self, off: int, n: int, copied) -> None:
        torch.cuda.current_stream().wait_event(copied)
        rows = self._gather(self.phase_a(n), self._rows)
        sig_rows = self._gather(self.phase_b1(n, rows), self._sig_rows)
        self.phase_b2(n, sig_rows)
        if self.world > 1:
            from .ingest import fetch_chunks_routed
            counts, slots = self.requests()
            ghost, lens = fetch_chunks_routed(torch.tensor(counts, dtype=torch.int64), slots, self.data, self._cuts, self._uniq, self.group)
            self.write_ghost(ghost, lens)
        self.phase_b3(n)

    # ------------------------------------------------------------------ results
    def read_state(self, check: bool = True) -> list:
        st = self._state.tolist()
        status = st[7]
        if check and self.world > 1 and self.group is not False:
            import torch.distributed as dist
            if dist.is_initialized():
                t = torch.tensor([status], dtype=torch.int64, device="cpu" if dist.get_backend(self.group) == "gloo" else self.dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
                status = max(status, int(t.item()))
        if check and status:
            raise ValueError(f"streaming chain status {status:#x} on some rank: bit0 chunk capacity, bit1 stored-chunk capacity, bit2 L2, bit3 exchange row, "
                             "bit4 workspace not initialised, bit5 state block inconsistent, bit6 a piece refused by push()"
                             + (f" (this rank: {self._host_error})" if self._host_error else "")
                             + ", bit7 more new stored chunks than a signature row holds, bits 8.. DEFLATE; the failing batch and every later one were dropped")
        return st

    def finish(self, check: bool = True) -> ShardResult:
        while self.pending:
            self._process(*self.pending.pop(0))
        st = self.read_state(check)
        n_done, n_c, nu, s_bytes, n_global = st[0], st[1], st[3], st[5], st[8]
        gidx = self._gidx[:n_c]
        base, bg, kind = self._base[:nu], self._base_global[:nu], self._kind[:nu]
        res = ShardResult(n_done, self._cuts[: n_c + 1], self._digests_g[gidx], 0, n_global, self._first_occ_g[gidx], self._refcount_g[gidx],
                          self._uniq[:nu], self._sig[:nu], self._band_keys[:nu], base, self._streams[:s_bytes], self._stream_off[: nu + 1], kind,
                          base_global=bg)
        res.gidx, res.ug = gidx, self._ug[:nu]
        import numpy as np
        from .manifest import REMOTE_BASE_DTYPE
        slots = ((bg >= 0) & (base < 0) & (kind == 2)).nonzero().flatten()
        tab = np.zeros(int(slots.numel()), REMOTE_BASE_DTYPE)
        if slots.numel():
            tab["slot"] = slots.cpu().numpy(); tab["shard"] = self._g_owner[bg[slots]].cpu().numpy(); tab["base_slot"] = self._g_local[bg[slots]].cpu().numpy()
        res.remote_bases = tab if len(tab) else None
        res.stats = shard_stats(res)
        return res


def stream_shards_local_gl4_graph(batches: list, cfg: IngestConfig, world: int, device, **kw) -> list:
    """A `world`-rank captured global-L4 stream with every rank on THIS GPU, in lock step: per global batch each phase of every rank, the
    exchanges delivered as the collectives would (rows concatenated in rank order; remote dictionaries served by the owner's `serve`).
    Each result is what the rank would hold after GraphGlobalL4StreamIngest.finish()."""
    seg = cfg.seg_size
    bounds = [deal_batch(b.numel(), world, seg) for b in batches]
    pb = -(-max(max(bd[r + 1] - bd[r] for r in range(world)) for bd in bounds) // seg) * seg
    local_total = [sum(bd[r + 1] - bd[r] for bd in bounds) for r in range(world)]
    ranks = [GraphGlobalL4StreamIngest(cfg, max(local_total[r], 1), pb, device, world, r, group=False, **kw) for r in range(world)]
    for b, bd in zip(batches, bounds):
        ns = []
        for r, s in enumerate(ranks):
            n = bd[r + 1] - bd[r]
            if n:
                s.data[s.n_bytes: s.n_bytes + n].copy_(b[bd[r]: bd[r + 1]])
            ns.append(n)
        rows = torch.cat([s.phase_a(n).clone() for n, s in zip(ns, ranks)])
        sig_rows = torch.cat([s.phase_b1(n, rows).clone() for n, s in zip(ns, ranks)])
        for n, s in zip(ns, ranks):
            s.phase_b2(n, sig_rows)
        for s in ranks:
            counts, slots = s.requests()
            parts, lens, o = [], [], 0
            for owner, cnt in enumerate(counts):
                if cnt:
                    by, ln = ranks[owner].serve(slots[o: o + cnt])
                    parts.append(by); lens.append(ln); o += cnt
            s.write_ghost(torch.cat(parts) if parts else torch.empty(0, dtype=torch.uint8, device=device),
                          torch.cat(lens) if lens else torch.empty(0, dtype=torch.int64, device=device))
        for n, s in zip(ns, ranks):
            s.phase_b3(n)
            s.n_bytes += n
    return [s.finish() for s in ranks]
