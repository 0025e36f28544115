"""Multi-GPU streaming (BASELINE.json configs[4]: "4 x 10 GB mixed corpora streamed, 8 x MI355X, hipGraph-captured per-batch
pipeline"; the reference's batch loop README.md:1519-1580 against ONE index, one process per GPU).

Definition.  The stream arrives as GLOBAL BATCHES.  A batch is dealt to the ranks as contiguous runs of whole segments
("pieces": rank 0 gets the first run, rank 1 the next, ...), so the concatenation of the pieces in (batch, rank) order IS the
logical stream, and the global chunk order is (batch, rank, local index) — the natural order of that stream.  Per batch:

  phase A   every rank: L2 FastCDC + L3 SHA-256 of ITS piece -> its exchange row {count, digests}       hmse_stream_piece_hash
  exchange  ONE all-gather of the rows (fixed-size rows: no count round trip, capturable)                RCCL over xGMI
  phase B   every rank: rows -> the global digest array in (batch, rank, local) order -> the same persistent L3 table on every
            rank (order-independent first occurrences, hmse_l3_index_update's rule) -> ITS stored chunks -> L4 MinHash +
            ITS persistent band tables -> L1 dictionary DEFLATE -> index tails, state advanced           hmse_stream_piece_encode

Dedupe is therefore GLOBAL over the stream so far (a chunk is stored by the rank that holds its first occurrence in stream
order); L4 bases and dictionaries are scoped to the rank (the dictionary bytes must be resident), exactly as in the sharded
one-shot ingest (ingest.ingest_shard(distributed=True)) — which is the one-batch special case of this definition.  No host
read happens between the stages; with graph=True phase A and phase B of a piece size are captured into two hipGraphs at their
second use and replayed from then on (the collective stays an ordinary stream-ordered RCCL call between the two replays).

`exchange(row) -> rows` replaces the collective: tests/ and ingest-on-one-GPU emulations drive several ranks in lock step
(stream_shards_local).  The oracle of this definition is tests/test_gpu_stream_dist.py::oracle_stream_pipeline.
"""
from __future__ import annotations

import torch

from . import ops
from .config import LAYER_L1, LAYER_L2, LAYER_L3, LAYER_L4, IngestConfig
from .ingest import ShardResult, shard_stats


import os as _os
_DEBUG_SYNC = _os.environ.get("HMSE_STREAM_DEBUG_SYNC") == "1"


def deal_batch(batch_bytes: int, world: int, seg_size: int) -> list:
    """Piece boundaries of a global batch of `batch_bytes`: contiguous runs of whole segments, rank r gets segments
    [r S / R, (r + 1) S / R) of the batch's S segments (the last segment may be partial: only the stream's last batch)."""
    n_seg = -(-batch_bytes // seg_size)
    b = [min(batch_bytes, (r * n_seg // world) * seg_size) for r in range(world)] + [batch_bytes]
    return b


def all_gather_rows(row: torch.Tensor, world: int, group=None, out: torch.Tensor | None = None) -> torch.Tensor:
    """THE collective of a multi-rank stream's batch: all-gather of the ranks' fixed-size exchange rows, in rank order.
    RCCL over xGMI: device to device, stream-ordered, no host read (1 GiB pieces: 16.8 MB per rank and batch).  gloo (the CPU
    tests and the one-GPU rehearsal): staged through host memory."""
    import torch.distributed as dist
    if out is None:
        out = torch.empty(world * row.numel(), dtype=torch.uint8, device=row.device)
    if dist.get_backend(group) == "gloo":
        rows = torch.empty(world * row.numel(), dtype=torch.uint8)
        dist.all_gather_into_tensor(rows, row.cpu(), group=group)
        out.copy_(rows)
    else:
        dist.all_gather_into_tensor(out, row, group=group)
    return out


class DistStreamIngest:
    """This rank's side of a stream sharded over `world` ranks.  Every rank pushes ITS piece of every global batch, in the
    same order (a rank with no bytes in a batch pushes an empty tensor) — the per-batch exchange is a collective."""

    def __init__(self, cfg: IngestConfig, capacity_bytes: int, piece_bytes: int, device, world: int, rank: int,
                 max_chunks_global: int | None = None, max_chunks: int | None = None, stream_capacity: int | None = None,
                 graph: bool = True, group=None, exchange=None, always_exchange: bool = False):
        if cfg.layers != (LAYER_L1 | LAYER_L2 | LAYER_L3 | LAYER_L4):
            raise ValueError("DistStreamIngest runs the full L1-L4 pipeline")
        if piece_bytes <= 0 or piece_bytes % cfg.seg_size:
            raise ValueError("piece_bytes (the nominal piece size) must be a positive multiple of seg_size")
        if not 0 <= rank < world <= 256:
            raise ValueError("0 <= rank < world <= 256")
        self.cfg, self.dev, self.world, self.rank, self.group = cfg, device, int(world), int(rank), group
        self.cap_bytes = int(piece_bytes)
        self.data = torch.empty(int(capacity_bytes), dtype=torch.uint8, device=device)
        self.n_bytes = 0
        self.copy_stream = torch.cuda.Stream(device=device)
        self.pending: list[tuple[int, int, torch.cuda.Event]] = []
        per = max(1, cfg.avg_size // 2)
        self.max_chunks = int(max_chunks or (capacity_bytes // per + capacity_bytes // cfg.seg_size + 64))
        self.max_chunks_g = int(max_chunks_global or self.max_chunks * world)
        self.max_unique = min(self.max_chunks, 1 << 23)
        mc, mg, mu = self.max_chunks, self.max_chunks_g, self.max_unique
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=device)
        # this rank's chunks
        self._cuts = z(mc + 1, torch.int64)
        self._gidx = z(mc, torch.int64)                       # local chunk -> global chunk index
        self._uniq = z(mu, torch.int64)                       # local ids of the chunks this rank stores, ascending
        self._sig = torch.empty((mu, cfg.n_hashes), dtype=torch.int32, device=device)
        self._band_keys = z((mu, cfg.bands), torch.int32)
        self._base = z(mu, torch.int64)
        self._kind = z(mu, torch.uint8)
        self._stream_off = z(mu + 1, torch.int64)
        self._lsh_tables = torch.empty((cfg.bands, ops.l4_lsh_slots(mu)), dtype=torch.int32, device=device)
        self._streams = torch.empty(int(stream_capacity or (capacity_bytes // 2 + (64 << 20))), dtype=torch.uint8, device=device)
        # the global index: identical on every rank
        self._digests_g = torch.empty((mg, 32), dtype=torch.uint8, device=device)
        self._first_occ_g = z(mg, torch.int64)
        self._refcount_g = z(mg, torch.int32)
        self._l3_table = torch.empty(ops.l3_index_slots(mg), dtype=torch.int32, device=device)
        ops.l3_index_update(self._digests_g, 0, 0, self._first_occ_g, self._refcount_g, self._l3_table)     # clears the table
        ops.l4_lsh_update(self._sig, 0, 0, cfg, self._band_keys, self._base, self._lsh_tables)               # clears the tables
        self._state = torch.zeros(16, dtype=torch.int64, device=device)
        self._host_error = None
        self._ws = ops.stream_workspace(self.cap_bytes, cfg, device)
        self.row_bytes = ops.stream_row_bytes(self.cap_bytes, cfg)
        self._row = torch.zeros(self.row_bytes, dtype=torch.uint8, device=device)
        self._rows = torch.zeros(self.world * self.row_bytes, dtype=torch.uint8, device=device)
        self.graph = bool(graph)
        self._graphs: dict[int, list] = {}      # piece bytes -> [seg_off, graph A or None, graph B or None, uses]
        self._exchange = exchange
        self._always_exchange = bool(always_exchange)    # world size 1 still runs the collective (hardware rehearsal of the graph / RCCL interleaving)
        self.n_batches = 0

    # ------------------------------------------------------------------ the exchange step
    def _all_gather(self, row: torch.Tensor) -> torch.Tensor:
        if self._exchange is not None:
            return self._exchange(row)
        if self.world == 1 and not self._always_exchange:
            return row
        return all_gather_rows(row, self.world, self.group, out=self._rows)

    # ------------------------------------------------------------------ the two phases (enqueue only)
    def _entry(self, n: int) -> list:
        e = self._graphs.get(n)
        if e is None:
            e = [ops.segment_offsets(n, self.cfg.seg_size, self.dev) if n else None, None, None, 0]
            self._graphs[n] = e
        return e

    def _call_hash(self, n: int, seg_off) -> None:
        ops.stream_piece_hash(self.data, n, self.cap_bytes, seg_off, self.cfg, self._state, self._cuts, self.max_chunks, self._row, self._ws)

    def _call_encode(self, n: int, rows: torch.Tensor) -> None:
        ops.stream_piece_encode(self.data, n, self.cap_bytes, self.cfg, self._state, rows, self.world, self.rank, self._cuts, self._gidx,
                                self._digests_g, self.max_chunks_g, self._first_occ_g, self._refcount_g, self._l3_table, self._uniq,
                                self.max_unique, self._sig, self._band_keys, self._base, self._lsh_tables, self._kind, self._stream_off,
                                self._streams, self._ws)

    def _run(self, which: int, n: int, fn) -> None:
        """First use of a piece size: plain enqueue (also sets the kernels' attributes before any capture); second use:
        capture into a hipGraph; from then on: replay."""
        e = self._entry(n)
        if _DEBUG_SYNC:     # HMSE_STREAM_DEBUG_SYNC=1: localise a device fault to (rank, batch, phase) — diagnostics only
            import sys
            torch.cuda.synchronize()
            cap = (self.row_bytes - 32) // 32
            o = (((cap + 1) * 8 + 255) // 256) * 256
            l2o = o + 512 + 2 * ((cap * 8 + 255) // 256 * 256) + 256 + (((cap + 1) * 8 + 255) // 256) * 256 + ((cap + 255) // 256) * 256 + 256 + ((self.row_bytes + 255) // 256) * 256
            print(f"[stream_dist] rank {self.rank} batch {self.n_batches} phase {'AB'[which - 1]} n={n} state={self._state.tolist()[:11]} n_cuts={self._ws[o:o + 8].view(torch.int64).item()} "
                  f"l2_status={self._ws[o + 256:o + 260].view(torch.int32).item()} l2hdr={self._ws[l2o:l2o + 32].view(torch.int64).tolist()}", file=sys.stderr, flush=True)
        if not self.graph or e[3] < 1:
            fn()
        else:
            if e[which] is None:
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    fn()
                e[which] = g
            e[which].replay()

    def hash_piece(self, n: int) -> torch.Tensor:
        """Phase A for the next piece (its bytes are at data[state.off ..)): returns this rank's exchange row."""
        e = self._entry(n)
        self._run(1, n, lambda: self._call_hash(n, e[0]))
        return self._row

    def encode_piece(self, n: int, rows: torch.Tensor) -> None:
        """Phase B: `rows` = the rows of all ranks in rank order (a tensor whose address is stable across batches when graphs are on)."""
        if rows.data_ptr() != self._rows.data_ptr() and not (self.world == 1 and rows.data_ptr() == self._row.data_ptr()):
            self._rows.copy_(rows.reshape(-1))
            rows = self._rows
        self._run(2, n, lambda: self._call_encode(n, rows))
        self._entry(n)[3] += 1
        self.n_batches += 1

    # ------------------------------------------------------------------ feeding
    def push(self, host_piece: torch.Tensor) -> None:
        """Issue the host -> HBM copy of this rank's piece of the next global batch, then process the piece whose copy was
        issued by the previous push (its kernels overlap this copy).  COLLECTIVE: every rank pushes once per global batch."""
        n = host_piece.numel()
        err = None
        if n > self.cap_bytes:
            err = "a piece may not exceed the stream's nominal piece size"
        elif self.n_bytes % self.cfg.seg_size:
            err = "only a rank's last piece may end inside a segment"
        elif self.n_bytes + n > self.data.numel():
            err = "stream capacity exceeded"
        if err:
            # A rank that raised here alone would leave its peers blocked in this batch's all-gather.  Instead the piece
            # is refused THROUGH the chain: sticky status bit 6 on the device (phase A then reports 0 chunks, every later batch of
            # this rank is a no-op), the rank keeps taking part in the collectives with empty rows, and read_state()'s
            # all-reduce(MAX) of the status makes EVERY rank raise at finish() — this one with the reason.
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
            self._state[7:8] |= 64      # (behind the piece pushed before, which is still good)
        self.pending.append((self.n_bytes, n, ev))
        self.n_bytes += n

    def _process(self, off: int, n: int, copied: torch.cuda.Event) -> None:
        torch.cuda.current_stream().wait_event(copied)
        row = self.hash_piece(n)
        rows = self._all_gather(row)
        self.encode_piece(n, rows)

    # ------------------------------------------------------------------ results
    def read_state(self, check: bool = True) -> list:
        st = self._state.tolist()
        status = st[7]
        if check and self.world > 1 and self._exchange is None:
            # a failed rank must fail the whole stream: its chunks are missing from every rank's index
            import torch.distributed as dist
            t = torch.tensor([status], dtype=torch.int64, device="cpu" if dist.get_backend(self.group) == "gloo" else self.dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            status = max(status, int(t.item()))
        if check and status:
            raise ValueError(f"streaming chain status {status:#x} on some rank: bit0 chunk capacity, bit1 stored-chunk capacity, bit2 L2, bit3 "
                             "exchange row, bit4 workspace not initialised, bit5 state block inconsistent, bit6 a piece refused by push()"
                             + (f" (this rank: {self._host_error})" if self._host_error else "")
                             + ", bits 8.. DEFLATE (0x100 stream capacity, 0x200 workspace); the failing batch and every later one were dropped")
        return st

    def finish(self, check: bool = True) -> ShardResult:
        while self.pending:
            self._process(*self.pending.pop(0))
        st = self.read_state(check)
        n_done, n_chunks, n_unique, s_bytes, n_global = st[0], st[1], st[3], st[5], st[8]
        gidx = self._gidx[:n_chunks]
        res = ShardResult(n_done, self._cuts[: n_chunks + 1], self._digests_g[gidx], 0, n_global, self._first_occ_g[gidx], self._refcount_g[gidx],
                          self._uniq[:n_unique], self._sig[:n_unique], self._band_keys[:n_unique], self._base[:n_unique], self._streams[:s_bytes],
                          self._stream_off[: n_unique + 1], self._kind[:n_unique])
        res.gidx = gidx
        res.stats = shard_stats(res)
        return res


def stream_shards_local(batches: list, cfg: IngestConfig, world: int, device, piece_bytes: int | None = None, graph: bool = True,
                        **kw) -> list:
    """A `world`-rank stream with every rank on THIS GPU, in lock step: phase A of every rank, the rows concatenated as the
    all-gather would deliver them, phase B of every rank — per global batch.  `batches`: host uint8 tensors (the global
    batches, whole segments except the last).  Each result is what the rank would hold after DistStreamIngest.finish()."""
    seg = cfg.seg_size
    bounds = [deal_batch(b.numel(), world, seg) for b in batches]
    pb = piece_bytes or max(max(bd[r + 1] - bd[r] for r in range(world)) for bd in bounds)
    pb = -(-pb // seg) * seg
    local_total = [sum(bd[r + 1] - bd[r] for bd in bounds) for r in range(world)]
    ranks = [DistStreamIngest(cfg, max(local_total[r], 1), pb, device, world, r, graph=graph, exchange=lambda row: row, **kw) for r in range(world)]
    rows = torch.zeros(world * ranks[0].row_bytes, dtype=torch.uint8, device=device)
    for b, bd in zip(batches, bounds):
        ns = []
        for r, s in enumerate(ranks):
            n = bd[r + 1] - bd[r]
            if n:
                s.data[s.n_bytes: s.n_bytes + n].copy_(b[bd[r]: bd[r + 1]])
            ns.append(n)
            row = s.hash_piece(n)
            rows[r * s.row_bytes: (r + 1) * s.row_bytes].copy_(row)
        for n, s in zip(ns, ranks):
            s.encode_piece(n, rows)
            s.n_bytes += n
    return [s.finish() for s in ranks]


def store_results(results: list, only: int | None = None) -> list:
    """The ranks' results of a multi-rank stream in the STORE numbering — what manifest.build_manifest needs to write one manifest
    per rank that manifest.merge_manifests can join: chunk indices become (shard, local) = vbase[shard] + local, so every shard's
    chunks are contiguous again (`chunk_base`, `shard_bases`), `first_occ` is translated, and `pieces` records where the shard's
    chunks sit in the STREAM order (the shards interleave; the reader restores the stream from it).  A first occurrence may live on
    a later-numbered shard (HMSE_MANIFEST_ANY_SHARD_TARGET).  `results`: every rank's DistStreamIngest.finish(), in rank order
    (stream_shards_local returns exactly that; N processes all-gather their `gidx` first — gather_global_index — and convert
    `only` their own)."""
    import dataclasses

    import numpy as np
    from .manifest import PIECE_DTYPE
    dev = results[0].cuts.device
    counts = [int(r.gidx.numel()) for r in results]
    vbase = [sum(counts[:i]) for i in range(len(results))]
    n_global = results[0].n_global
    to_store = torch.full((n_global,), -1, dtype=torch.int64, device=dev)
    for r, vb in zip(results, vbase):
        to_store[r.gidx] = torch.arange(vb, vb + r.gidx.numel(), dtype=torch.int64, device=dev)
    if bool((to_store < 0).any()):
        raise ValueError("store_results: the ranks' chunks do not tile the stream")
    out = []
    for rank, (r, vb) in enumerate(zip(results, vbase)):
        if only is not None and rank != only:
            out.append(None)
            continue
        g = r.gidx.cpu().numpy()
        starts = np.nonzero(np.diff(g, prepend=g[:1] - 2) != 1)[0] if len(g) else np.zeros(0, np.int64)
        ends = np.append(starts[1:], len(g))
        pieces = np.zeros(len(starts), PIECE_DTYPE)
        pieces["g0"] = g[starts] if len(g) else 0
        pieces["n"] = ends - starts
        rr = dataclasses.replace(r, first_occ=to_store[r.first_occ], chunk_base=vb, shard_bases=list(vbase))
        rr.gidx, rr.pieces = r.gidx, pieces
        for extra in ("ug", "remote_bases"):          # a global-L4 stream's shard (GlobalL4StreamIngest)
            if hasattr(r, extra):
                setattr(rr, extra, getattr(r, extra))
        out.append(rr)
    return out


def gather_global_index(res, group=None):
    """N processes: every rank's `gidx` (all-gather; 8 B per chunk), as the list store_results() wants — only the index maps travel,
    the other ranks' results stay where they are: returns light stand-ins carrying (gidx, n_global) for the other ranks."""
    import types
    import torch.distributed as dist
    from .ingest import gather_rows
    allg, base, total, bases = gather_rows(res.gidx.reshape(-1, 1), group)
    world = dist.get_world_size(group)
    bases = list(bases) + [total]
    out = []
    for r in range(world):
        out.append(res if r == dist.get_rank(group) else types.SimpleNamespace(gidx=allg[bases[r]: bases[r + 1], 0], n_global=res.n_global, cuts=res.cuts))
    return out


# ---------------------------------------------------------------------------------------------------------------------------
# Global L4 for a multi-rank stream ("cross-GPU base-chunk fetch over xGMI for global L4 (config 5)")
# ---------------------------------------------------------------------------------------------------------------------------
class GlobalL4StreamIngest:
    """This rank's side of a multi-rank stream whose L4 base selection spans ALL ranks (README.md:1375-1383 against one band
    table): besides the digests, every batch all-gathers the signatures of its newly stored chunks; every rank keeps the same
    band tables over the GLOBAL stored-chunk order (batch, rank, local) and so finds the same bases as a one-rank run; a base
    stored on another rank is fetched over xGMI (three all-to-alls, ingest.fetch_chunks_routed) into a ghost area behind the
    rank's data, where the DEFLATE kernel addresses it by chunk id.  Dedupe AND dictionaries are then those of the one-rank
    stream: the records of all ranks, put in global stored order, are bit-identical to its records.

    Per batch the host sizes three ragged exchanges, so this path is enqueued stage by stage (no hipGraph); the shard-local
    DistStreamIngest stays the captured fast path.  The stages are methods so that tests/ can drive several ranks in lock step on
    one GPU (stream_shards_local_global_l4): stage_hash -> [digests of all ranks] -> stage_index -> [signatures of all ranks] ->
    stage_lsh -> [ghost chunks] -> stage_encode.

    finish() -> ShardResult with: gidx (local chunk -> global chunk), ug (local stored slot -> global stored index), base (local
    slot of the dictionary, -1 if none or remote), base_global (GLOBAL STORED index of the dictionary in stream order, -1 none),
    remote_bases ((slot, owner rank, owner's slot) table for manifest.build_manifest)."""

    def __init__(self, cfg: IngestConfig, capacity_bytes: int, piece_bytes: int, device, world: int, rank: int, group=None,
                 max_chunks: int | None = None, max_chunks_global: int | None = None, ghost_bytes: int | None = None):
        if cfg.layers != (LAYER_L1 | LAYER_L2 | LAYER_L3 | LAYER_L4):
            raise ValueError("GlobalL4StreamIngest runs the full L1-L4 pipeline")
        if piece_bytes <= 0 or piece_bytes % cfg.seg_size:
            raise ValueError("piece_bytes (the nominal piece size) must be a positive multiple of seg_size")
        if not 0 <= rank < world <= 256:
            raise ValueError("0 <= rank < world <= 256")
        self.cfg, self.dev, self.world, self.rank, self.group = cfg, device, int(world), int(rank), group
        self.capacity = int(capacity_bytes)
        self.ghost_cap = int(ghost_bytes if ghost_bytes is not None else 2 * piece_bytes)
        self.data = torch.empty(self.capacity + self.ghost_cap, dtype=torch.uint8, device=device)   # [0, capacity): this rank's pieces; behind: ghosts
        self.n_bytes = 0
        self.copy_stream = torch.cuda.Stream(device=device)
        self.pending: list = []
        per = max(1, cfg.avg_size // 2)
        self.max_chunks = int(max_chunks or (capacity_bytes // per + capacity_bytes // cfg.seg_size + 64))
        self.max_chunks_g = int(max_chunks_global or self.max_chunks * world)
        mc, mg = self.max_chunks, self.max_chunks_g
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=device)
        self._cuts = z(mc + 1, torch.int64)
        self._gidx = z(mc, torch.int64)
        self.n_chunks = 0
        # this rank's stored chunks (appended per batch)
        self._uniq, self._ug, self._base, self._base_global, self._kind, self._sig_l, self._keys_l = [], [], [], [], [], [], []
        self._stream_parts, self._stream_lens = [], []
        self.n_unique = 0
        # the global index and the global band tables: identical on every rank
        self._digests_g = torch.empty((mg, 32), dtype=torch.uint8, device=device)
        self._first_occ_g = z(mg, torch.int64)
        self._refcount_g = z(mg, torch.int32)
        self._l3_table = torch.empty(ops.l3_index_slots(mg), dtype=torch.int32, device=device)
        self._mug = min(mg, 1 << 23)
        self._sig_g = torch.empty((self._mug, cfg.n_hashes), dtype=torch.int32, device=device)
        self._keys_g = z((self._mug, cfg.bands), torch.int32)
        self._base_g = z(self._mug, torch.int64)
        self._lsh_tables = torch.empty((cfg.bands, ops.l4_lsh_slots(self._mug)), dtype=torch.int32, device=device)
        self._g_owner = z(self._mug, torch.int64)     # global stored chunk -> owning rank, that rank's stored slot
        self._g_local = z(self._mug, torch.int64)
        ops.l3_index_update(self._digests_g, 0, 0, self._first_occ_g, self._refcount_g, self._l3_table)
        ops.l4_lsh_update(self._sig_g, 0, 0, cfg, self._keys_g, self._base_g, self._lsh_tables)
        self.n_global = 0
        self.u_global = 0
        self.u_counts = [0] * self.world
        self._ws = None
        self.n_batches = 0
        self.remote_dictionaries = 0
        self.ghost_bytes_fetched = 0

    # ------------------------------------------------------------------ the four stages of one batch
    def stage_hash(self, off: int, n: int) -> torch.Tensor:
        """L2 + L3 of this rank's piece (bytes [off, off + n) of its data): cuts appended, -> digests uint8[n_b, 32]."""
        n_old = self.n_chunks
        if n:
            cuts_b = ops.l2_cdc(self.data[off: off + n], self.cfg)
            n_b = cuts_b.numel() - 1
            if n_old + n_b > self.max_chunks:
                raise ValueError("stream index capacity exceeded (max_chunks)")
            self._cuts[n_old + 1: n_old + n_b + 1] = cuts_b[1:] + off
            dg = ops.l3_sha256(self.data[: off + n], self._cuts[n_old: n_old + n_b + 1])
        else:
            n_b, dg = 0, torch.empty((0, 32), dtype=torch.uint8, device=self.dev)
        self._cur = dict(off=off, n=n, n_old=n_old, n_b=n_b)
        return dg

    def stage_index(self, all_digests: torch.Tensor, counts: list) -> torch.Tensor:
        """The batch's digests of ALL ranks (rank order) join the global index; -> signatures int32[u_b, 128] of the chunks THIS
        rank stores (first occurrences in global order)."""
        c = self._cur
        g_old, g_new = self.n_global, int(sum(counts))
        if g_old + g_new > self.max_chunks_g:
            raise ValueError("global index capacity exceeded (max_chunks_global)")
        self._digests_g[g_old: g_old + g_new] = all_digests
        ops.l3_index_update(self._digests_g, g_old, g_new, self._first_occ_g, self._refcount_g, self._l3_table)
        mine0 = g_old + int(sum(counts[: self.rank]))
        g = torch.arange(mine0, mine0 + c["n_b"], dtype=torch.int64, device=self.dev)
        self._gidx[c["n_old"]: c["n_old"] + c["n_b"]] = g
        uniq_new = c["n_old"] + (self._first_occ_g[mine0: mine0 + c["n_b"]] == g).nonzero().flatten()
        self.n_global = g_old + g_new
        self.n_chunks = c["n_old"] + c["n_b"]
        c["uniq_new"] = uniq_new
        end = c["off"] + c["n"]
        c["sig"] = ops.l4_minhash(self.data[:end], self._cuts[: self.n_chunks + 1], self.cfg, uniq_new) if uniq_new.numel() else \
            torch.empty((0, self.cfg.n_hashes), dtype=torch.int32, device=self.dev)
        return c["sig"]

    def stage_lsh(self, all_sigs: torch.Tensor, counts_u: list):
        """The batch's signatures of ALL ranks (rank order) join the global band tables; this rank's bases are resolved.
        -> (counts per owner rank, owners' local stored slots grouped by owner): the remote dictionaries to fetch."""
        c = self._cur
        u_old, u_new = self.u_global, int(sum(counts_u))
        if u_old + u_new > self._mug:
            raise ValueError("global stored-chunk capacity exceeded")
        if u_new:
            self._sig_g[u_old: u_old + u_new] = all_sigs
            o = u_old
            for r, cnt in enumerate(counts_u):
                if cnt:
                    self._g_owner[o: o + cnt] = r
                    self._g_local[o: o + cnt] = torch.arange(self.u_counts[r], self.u_counts[r] + cnt, dtype=torch.int64, device=self.dev)
                    self.u_counts[r] += cnt
                    o += cnt
            ops.l4_lsh_update(self._sig_g, u_old, u_new, self.cfg, self._keys_g, self._base_g, self._lsh_tables)
        self.u_global = u_old + u_new
        m0 = u_old + int(sum(counts_u[: self.rank]))
        u_b = int(counts_u[self.rank])
        bg = self._base_g[m0: m0 + u_b].clone()
        has = bg >= 0
        own = has & (self._g_owner[bg.clamp(min=0)] == self.rank)
        remote = has & ~own
        c.update(m0=m0, u_b=u_b, bg=bg, own=own, remote=remote)
        req = torch.unique(bg[remote])                                     # ascending global stored indices
        owner = self._g_owner[req]
        perm = torch.argsort(owner, stable=True)                            # grouped by owner, ascending inside a group
        c["req"] = req[perm]
        counts = torch.bincount(owner, minlength=self.world)[: self.world]
        return counts, self._g_local[c["req"]]

    def serve(self, local_slots: torch.Tensor):
        """Raw bytes of this rank's stored chunks `local_slots` (what fetch_chunks_routed sends to a peer): (bytes, lens)."""
        uniq = torch.cat(self._uniq) if self._uniq else torch.empty(0, dtype=torch.int64, device=self.dev)
        c = getattr(self, "_cur", None)
        if c is not None and "uniq_new" in c and not c.get("committed", False):
            uniq = torch.cat([uniq, c["uniq_new"]])                          # (a chunk of THIS batch can already be a peer's dictionary)
        cid = uniq[local_slots]
        ln = self._cuts[cid + 1] - self._cuts[cid]
        oc = torch.zeros(cid.numel() + 1, dtype=torch.int64, device=self.dev)
        torch.cumsum(ln, 0, out=oc[1:])
        return (ops.read_assemble(oc, cid, self._cuts, self.data) if cid.numel() else torch.empty(0, dtype=torch.uint8, device=self.dev)), ln

    def stage_encode(self, ghost: torch.Tensor, ghost_lens: torch.Tensor) -> None:
        """DEFLATE of this rank's newly stored chunks; a remote dictionary is ghost chunk j (the j-th request of stage_lsh)."""
        c = self._cur
        cfg, dev = self.cfg, self.dev
        uniq_new, bg, own, remote = c["uniq_new"], c["bg"], c["own"], c["remote"]
        gb = int(ghost.numel())
        if gb > self.ghost_cap:
            raise ValueError(f"a batch needs {gb} bytes of remote dictionaries, the ghost area holds {self.ghost_cap} (ghost_bytes=)")
        uniq_all = torch.cat(self._uniq + [uniq_new]) if (self._uniq or uniq_new.numel()) else uniq_new
        base_local = torch.where(own, self._g_local[bg.clamp(min=0)], torch.full_like(bg, -1))
        base_chunk = torch.where(own, uniq_all[base_local.clamp(min=0)] if uniq_all.numel() else bg, torch.full_like(bg, -1))
        cuts_x = self._cuts[: self.n_chunks + 1]
        if c["req"].numel():
            self.data[self.capacity: self.capacity + gb] = ghost
            gcuts = self.capacity + torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(ghost_lens, 0)])
            cuts_x = torch.cat([cuts_x, gcuts])                              # chunk n_chunks = the gap up to the ghost area (never named)
            rs, inv = torch.sort(c["req"])
            j = inv[torch.searchsorted(rs, bg.clamp(min=0)).clamp(max=rs.numel() - 1)]
            base_chunk = torch.where(remote, self.n_chunks + 1 + j, base_chunk)
            self.remote_dictionaries += int(remote.sum().item())
            self.ghost_bytes_fetched += gb
        if uniq_new.numel():
            n_b = c["n_b"]
            worst = (5 * c["n"] + 1600 * n_b) + ops.workspace_bytes(ops.STAGE_DEFLATE, n_b, cfg) + (1 << 20)
            if self._ws is None or self._ws.numel() < worst:
                self._ws = None
                self._ws = torch.empty(worst, dtype=torch.uint8, device=dev)
            streams, off, kind = ops.l1_deflate(self.data, cuts_x, cfg, uniq_new, base_chunk, base_is_chunk_id=True, ws=self._ws)
            self._stream_parts.append(streams); self._stream_lens.append(off[1:] - off[:-1])
            self._kind.append(kind)
        self._uniq.append(uniq_new)
        self._ug.append(torch.arange(c["m0"], c["m0"] + c["u_b"], dtype=torch.int64, device=dev))
        self._base.append(base_local); self._base_global.append(bg)
        self._sig_l.append(c["sig"]); self._keys_l.append(self._keys_g[c["m0"]: c["m0"] + c["u_b"]].clone())
        self.n_unique += c["u_b"]
        c["committed"] = True
        self.n_batches += 1

    # ------------------------------------------------------------------ N processes: the stages joined by collectives
    def _agree(self, err) -> None:
        """COLLECTIVE: did this rank's last step fail?  If ANY rank says yes, EVERY rank raises — a rank that raised alone between two
        collectives would leave its peers blocked in the next one with no diagnosis.  (This mode is host-sized: it
        synchronises per stage anyway.)"""
        if self.world == 1:
            if err is not None:
                raise err
            return
        import torch.distributed as dist
        flags = [None] * self.world
        dist.all_gather_object(flags, None if err is None else str(err), group=self.group)
        bad = [(r, m) for r, m in enumerate(flags) if m is not None]
        if bad:
            self.failed = True
            raise ValueError("global-L4 stream abandoned on every rank: " + "; ".join(f"rank {r}: {m}" for r, m in bad))

    def _guard(self, fn, *args):
        """Run one stage; its rank-local refusal (a capacity) is agreed with the peers before anybody meets the next collective."""
        try:
            out, err = fn(*args), None
        except ValueError as e:
            out, err = None, e
        self._agree(err)
        return out

    def push(self, host_piece: torch.Tensor) -> None:
        """COLLECTIVE (every rank pushes once per global batch, an empty tensor if it has no bytes in it): issue the host -> HBM
        copy of this piece, then process the piece pushed before."""
        n = host_piece.numel()
        err = None
        if self.n_bytes % self.cfg.seg_size:
            err = ValueError("only a rank's last piece may end inside a segment")
        elif self.n_bytes + n > self.capacity:
            err = ValueError("stream capacity exceeded")
        self._agree(err)
        ev = torch.cuda.Event()
        with torch.cuda.stream(self.copy_stream):
            if n:
                self.data[self.n_bytes: self.n_bytes + n].copy_(host_piece, non_blocking=True)
            ev.record(self.copy_stream)
        while self.pending:
            self._process(*self.pending.pop(0))
        self.pending.append((self.n_bytes, n, ev))
        self.n_bytes += n

    def _process(self, off: int, n: int, copied) -> None:
        from .ingest import fetch_chunks_routed, gather_rows
        torch.cuda.current_stream().wait_event(copied)
        dg = self._guard(self.stage_hash, off, n)
        if self.world == 1:       # one rank: no process group needed, every dictionary is local
            sig = self.stage_index(dg, [int(dg.shape[0])])
            self.stage_lsh(sig, [int(sig.shape[0])])
            self.stage_encode(torch.empty(0, dtype=torch.uint8, device=self.dev), torch.empty(0, dtype=torch.int64, device=self.dev))
            return
        alld, _, _, bases = gather_rows(dg, self.group)
        counts = [b - a for a, b in zip(bases, list(bases[1:]) + [alld.shape[0]])]
        sig = self._guard(self.stage_index, alld, counts)
        alls, _, _, ub = gather_rows(sig, self.group)
        counts_u = [b - a for a, b in zip(ub, list(ub[1:]) + [alls.shape[0]])]
        rc, rl = self._guard(self.stage_lsh, alls, counts_u)
        uniq_all = torch.cat(self._uniq + [self._cur["uniq_new"]])
        ghost, glens = fetch_chunks_routed(rc, rl, self.data, self._cuts, uniq_all, self.group)
        self._guard(self.stage_encode, ghost, glens)

    def finish(self) -> ShardResult:
        while self.pending:
            self._process(*self.pending.pop(0))
        dev = self.dev
        cat = lambda parts, dt, shape=(0,): torch.cat(parts) if parts else torch.empty(shape, dtype=dt, device=dev)
        n_c = self.n_chunks
        gidx = self._gidx[:n_c]
        uniq = cat(self._uniq, torch.int64)
        lens = cat(self._stream_lens, torch.int64)
        off = torch.zeros(lens.numel() + 1, dtype=torch.int64, device=dev)
        torch.cumsum(lens, 0, out=off[1:])
        base, bg = cat(self._base, torch.int64), cat(self._base_global, torch.int64)
        kind = cat(self._kind, torch.uint8)
        res = ShardResult(self.n_bytes, self._cuts[: n_c + 1], self._digests_g[gidx], 0, self.n_global, self._first_occ_g[gidx], self._refcount_g[gidx],
                          uniq, cat(self._sig_l, torch.int32, (0, self.cfg.n_hashes)), cat(self._keys_l, torch.int32, (0, self.cfg.bands)), base,
                          cat(self._stream_parts, torch.uint8), off, kind, base_global=bg)
        res.gidx, res.ug = gidx, cat(self._ug, torch.int64)
        import numpy as np
        from .manifest import REMOTE_BASE_DTYPE
        slots = ((bg >= 0) & (base < 0) & (kind == 2)).nonzero().flatten()
        tab = np.zeros(int(slots.numel()), REMOTE_BASE_DTYPE)
        if slots.numel():
            tab["slot"] = slots.cpu().numpy(); tab["shard"] = self._g_owner[bg[slots]].cpu().numpy(); tab["base_slot"] = self._g_local[bg[slots]].cpu().numpy()
        res.remote_bases = tab if len(tab) else None
        res.stats = shard_stats(res)
        return res


def stream_shards_local_global_l4(batches: list, cfg: IngestConfig, world: int, device, **kw) -> list:
    """A `world`-rank global-L4 stream with every rank on THIS GPU, in lock step: per global batch each stage of every rank, the
    exchanges delivered as the collectives would (concatenation in rank order; remote dictionaries served by the owner's
    `serve`).  Each result is what the rank would hold after GlobalL4StreamIngest.finish()."""
    seg = cfg.seg_size
    bounds = [deal_batch(b.numel(), world, seg) for b in batches]
    pb = -(-max(max(bd[r + 1] - bd[r] for r in range(world)) for bd in bounds) // seg) * seg
    local_total = [sum(bd[r + 1] - bd[r] for bd in bounds) for r in range(world)]
    ranks = [GlobalL4StreamIngest(cfg, max(local_total[r], 1), pb, device, world, r, **kw) for r in range(world)]
    for b, bd in zip(batches, bounds):
        ns, dgs = [], []
        for r, s in enumerate(ranks):
            n = bd[r + 1] - bd[r]
            if n:
                s.data[s.n_bytes: s.n_bytes + n].copy_(b[bd[r]: bd[r + 1]])
            ns.append(n)
            dgs.append(s.stage_hash(s.n_bytes, n))
        alld, counts = torch.cat(dgs), [int(d.shape[0]) for d in dgs]
        sigs = [s.stage_index(alld, counts) for s in ranks]
        alls, counts_u = torch.cat(sigs), [int(x.shape[0]) for x in sigs]
        reqs = [s.stage_lsh(alls, counts_u) for s in ranks]
        for s, (rc, rl) in zip(ranks, reqs):
            parts, lens, o = [], [], 0
            for owner, cnt in enumerate(rc.tolist()):
                if cnt:
                    by, ln = ranks[owner].serve(rl[o: o + cnt])
                    parts.append(by); lens.append(ln); o += cnt
            ghost = torch.cat(parts) if parts else torch.empty(0, dtype=torch.uint8, device=device)
            glens = torch.cat(lens) if lens else torch.empty(0, dtype=torch.int64, device=device)
            s._ghost = (ghost, glens)
        for n, s in zip(ns, ranks):
            s.stage_encode(*s._ghost)
            s.n_bytes += n
    return [s.finish() for s in ranks]
