// stream_batch.hip — one batch of the streaming front end without a host read (BASELINE.json configs[4]
// "4 x 10 GB streamed, 8 x MI355X, hipGraph-captured per-batch pipeline"; the reference's batch loop README.md:1519-1580).
//
// hmse_amd/stream.py's eager path reads device counts on the host between stages (chunk count, stored-chunk count, stream
// bytes) to size the next launch.  Here every stage takes its ranges from a small state block in HBM (common.h, SB_*), grids
// and workspaces are sized for the worst case of the batch (bytes / min_size chunks), and the chain ends by advancing the
// state itself — so the sequence L2 -> L3 hash | digest exchange | L3 persistent index -> stored-chunk selection -> L4
// (MinHash + persistent band tables) -> L1 (dictionary DEFLATE) -> index tails can be captured into hipGraphs once and
// replayed for every batch; the host reads the state block when it wants the counts (at the end of the stream).
//
// The chain has TWO phases around the one exchange step of a multi-rank stream (one process per GPU):
//   hmse_stream_piece_hash    this rank's piece of the batch: L2 cut points, cuts appended, SHA-256 of the new chunks
//                             -> the rank's EXCHANGE ROW {u64 count, 24 B pad, cap x 32 B digests}
//   (all-gather of the rows over RCCL/xGMI — the only collective; with one rank the row IS the gathered array)
//   hmse_stream_piece_encode  rows of all ranks -> global digest array in (batch, rank, local) order, persistent L3 table
//                             (every rank runs the same order-independent first-occurrence rule on the same rows), this
//                             rank's stored chunks, L4, L1, index tails, state advanced
// hmse_stream_batch (one rank) runs both back to back.  Results equal the eager path's bit for bit (tests/test_gpu_stream.py),
// and a multi-rank stream equals the oracle's (n_ranks, piece_bytes) pipeline (tests/test_gpu_stream_dist.py).
// A batch that hits a capacity limit or a stage error sets the sticky state[SB_STATUS] and is dropped; every later batch of
// the stream is then a no-op (nothing is appended to an index that is no longer consistent), the host sees the status in finish().
#include "common.h"

namespace sb {

constexpr uint32_t ROW_HDR = 32;   // bytes in front of a row's digests: u64 chunk count + padding (keeps the digests 32-byte aligned)

// cuts of the piece (piece-local, from L2) -> this rank's cut array; N_NEW; the row's count.  n_cuts == nullptr: an empty piece.
__global__ __launch_bounds__(256) void append_cuts_kernel(const uint64_t* __restrict__ local, const uint64_t* __restrict__ n_cuts,
                                                           const uint32_t* __restrict__ l2_status, uint64_t* __restrict__ cuts_all,
                                                           uint64_t* st, uint64_t max_chunks, uint64_t cap, uint8_t* row) {
  const uint64_t n_old = st[SB_N_OLD], off = st[SB_OFF];
  uint64_t n_new = n_cuts ? *n_cuts : 0ull;
  uint64_t err = 0;
  if (st[SB_STATUS]) n_new = 0;                                      // an earlier batch failed: this one is a no-op
  else if (l2_status && *l2_status) { err = 4ull; n_new = 0; }       // the cut list is not trustworthy: drop the batch
  else if (n_new > cap || n_old + n_new > max_chunks) { err = 1ull; n_new = 0; }   // capacity: dropped and flagged, nothing out of bounds
  for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < n_new; j += (uint64_t)gridDim.x * 256) cuts_all[n_old + 1 + j] = local[j + 1] + off;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st[SB_N_NEW] = n_new;
    if (err) st[SB_STATUS] |= err;
    *(uint64_t*)row = n_new;
  }
}

// rows of all ranks -> digests_g[G_OLD + prefix(rank) + j]; G_NEW, G_BASE; local chunk -> global index.  One thread per
// (rank, chunk slot, 16-byte half); every thread derives the same counts from the row headers.
__global__ __launch_bounds__(256) void ingest_rows_kernel(const uint8_t* __restrict__ rows, uint64_t row_bytes, uint32_t world, uint32_t rank,
                                                           uint64_t cap, uint8_t* __restrict__ digests_g, uint64_t max_chunks_g,
                                                           uint64_t* __restrict__ gidx, uint64_t* st) {
  const uint64_t g_old = st[SB_G_OLD];
  const bool dead = st[SB_STATUS] != 0;                              // this rank is out (its own row says 0 chunks already)
  const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint32_t r = (uint32_t)(tid / (2 * cap));
  const uint64_t j = (tid % (2 * cap)) >> 1;
  const uint32_t half = (uint32_t)(tid & 1);
  uint64_t total = 0, before = 0, mine_before = 0, cnt_r = 0;
  bool bad = false;
  for (uint32_t q = 0; q < world; q++) {
    uint64_t c = *(const uint64_t*)(rows + (size_t)q * row_bytes);
    if (c > cap) { bad = true; c = 0; }                              // a row that does not come from this build's phase A
    if (q < r) before += c;
    if (q < rank) mine_before += c;
    if (q == r) cnt_r = c;
    total += c;
  }
  const bool over = g_old + total > max_chunks_g;
  // one rank: the global chunk order IS the local one, so the two counters must agree (a caller resuming a stream with the
  // ABI-1 state layout has state[8] = 0 and would overwrite digests from index 0 on) — refused with sticky status bit 5
  const bool stale = world == 1 && g_old != st[SB_N_OLD];
  if (dead || bad || over || stale) total = 0;
  if (r < world && total && j < cnt_r) {
    const uint4 v = *(const uint4*)(rows + (size_t)r * row_bytes + ROW_HDR + 32 * j + 16 * half);
    *(uint4*)(digests_g + 32 * (g_old + before + j) + 16 * half) = v;
  }
  if (gidx && total && r == rank && half == 0 && j < cnt_r) gidx[st[SB_N_OLD] + j] = g_old + mine_before + j;
  if (tid == 0) {
    st[SB_G_NEW] = total;
    st[SB_G_BASE] = g_old + mine_before;
    if (!dead && (bad || over || stale)) { st[SB_STATUS] |= stale ? 32ull : bad ? 8ull : 1ull; st[SB_N_NEW] = 0; }
  }
}

// stored chunks of the piece = this rank's chunks that are their own (global) first occurrence: ascending append of their
// LOCAL chunk ids to uniq_all, count -> state.  One workgroup: a piece has at most a few hundred thousand chunks.
__global__ __launch_bounds__(1024) void select_uniq_kernel(const uint64_t* __restrict__ first_occ_g, uint64_t* __restrict__ uniq_all,
                                                            uint64_t* st, uint64_t max_unique) {
  __shared__ uint32_t red[1024 / 64 + 1];
  __shared__ uint64_t run;
  const uint64_t n_old = st[SB_N_OLD], n_new = st[SB_N_NEW], u_old = st[SB_U_OLD], g_base = st[SB_G_BASE];
  if (threadIdx.x == 0) run = 0;
  __syncthreads();
  for (uint64_t b0 = 0; b0 < n_new; b0 += 1024) {
    const uint64_t j = b0 + threadIdx.x;
    const bool mine = j < n_new && first_occ_g[g_base + j] == g_base + j;
    uint32_t total;
    const uint32_t ex = block_exclusive_scan<1024>(mine ? 1u : 0u, red, &total);
    const uint64_t r = run;
    if (mine && u_old + r + ex < max_unique) uniq_all[u_old + r + ex] = n_old + j;
    __syncthreads();
    if (threadIdx.x == 0) run = r + total;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    uint64_t nu = run;
    if (u_old + nu > max_unique) { nu = 0; st[SB_STATUS] |= 2ull; }
    st[SB_U_NEW] = nu;
  }
}

// the piece's DEFLATE selection as contiguous arrays: chunk ids of its stored chunks and their dictionaries as chunk ids
__global__ __launch_bounds__(256) void prep_deflate_kernel(const uint64_t* __restrict__ uniq_all, const int64_t* __restrict__ base_all,
                                                            const uint64_t* __restrict__ st, uint64_t* __restrict__ sel_ids,
                                                            int64_t* __restrict__ sel_base, uint64_t* __restrict__ n_sel) {
  const uint64_t u_old = st[SB_U_OLD], nu = st[SB_U_NEW];
  const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (k == 0) *n_sel = nu;
  if (k >= nu) return;
  sel_ids[k] = uniq_all[u_old + k];
  const int64_t b = base_all[u_old + k];
  sel_base[k] = b >= 0 ? (int64_t)uniq_all[b] : -1;
}

// tails of kind / stream offsets
__global__ __launch_bounds__(256) void commit_kernel(const uint8_t* __restrict__ kind_b, const uint64_t* __restrict__ out_off_b, uint64_t cap_sel,
                                                      uint8_t* __restrict__ kind_all, uint64_t* __restrict__ stream_off_all,
                                                      const uint32_t* __restrict__ dfl_status, uint64_t* st) {
  const uint64_t u_old = st[SB_U_OLD], nu = st[SB_U_NEW], s_old = st[SB_S_OLD];
  const uint64_t total = nu ? out_off_b[cap_sel] : 0ull;
  for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k < nu; k += (uint64_t)gridDim.x * 256) {
    kind_all[u_old + k] = kind_b[k];
    stream_off_all[u_old + k + 1] = (k + 1 < nu ? out_off_b[k + 1] : total) + s_old;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st[SB_S_NEW] = total;
    if (*dfl_status) st[SB_STATUS] |= (uint64_t)*dfl_status << 8;
  }
}
// the state moves on to the next batch — unless this one (or an earlier one) failed: then the counters stay where the last
// good batch left them
__global__ void advance_kernel(uint64_t* st, uint64_t piece_bytes) {
  if (st[SB_STATUS]) return;
  st[SB_OFF] += piece_bytes;
  st[SB_N_OLD] += st[SB_N_NEW];
  st[SB_U_OLD] += st[SB_U_NEW];
  st[SB_S_OLD] += st[SB_S_NEW];
  st[SB_G_OLD] += st[SB_G_NEW];
}

// ---- global L4 for a multi-rank stream as captured phases (round 4; SURVEY.md §8f-3 "cross-GPU base-chunk fetch ... for global L4") ----
// The stream's stored chunks of ALL ranks are numbered in global stored order (batch, rank, local); every rank keeps the same
// signature array, band tables and owner map of that numbering, so every rank finds the same dictionary for every chunk as ONE rank
// ingesting the whole stream would.  Per batch, between the digest exchange and DEFLATE:
//   hmse_stream_piece_sign    rows of all ranks -> global index -> this rank's new stored chunks -> MinHash -> the rank's SIGNATURE ROW
//                             {u64 count, u64 first local stored slot, 16 B pad, sig_cap x 512 B}: fixed size, like the digest row
//   (all-gather of the signature rows)
//   hmse_stream_piece_bases   rows -> global signature array / owner map -> global band tables -> dictionary of each of this rank's new
//                             chunks: own (a chunk id), or REMOTE -> a request (owner, owner's stored slot), grouped by owner
//   (the remote dictionaries are fetched — the one step that stays eager: its all-to-alls are sized by the request counts)
//   hmse_stream_piece_encode_g  DEFLATE (a fetched dictionary is chunk ghost_chunk0 + j of the cut array) -> tails -> both states advance
constexpr uint32_t SIG_HDR = 32;
constexpr uint32_t SIG_BYTES = 512;   // 128 x u32 (the device path is specialised to n_hashes = 128)

// this rank's new signatures (sig_all[u_old .. u_old + u_new)) -> its exchange row; a batch with more new stored chunks than a row holds
// is dropped with status bit 7
__global__ __launch_bounds__(256) void sig_row_kernel(const uint32_t* __restrict__ sig_all, uint64_t* st, uint64_t sig_cap, uint8_t* __restrict__ row) {
  const uint64_t u_old = st[SB_U_OLD];
  uint64_t u_new = st[SB_U_NEW];
  const bool over = u_new > sig_cap;
  if (over || st[SB_STATUS]) u_new = 0;
  const uint4* src = (const uint4*)(sig_all + 128 * u_old);
  uint4* dst = (uint4*)(row + SIG_HDR);
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < u_new * (SIG_BYTES / 16); i += (uint64_t)gridDim.x * 256) dst[i] = src[i];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ((uint64_t*)row)[0] = u_new; ((uint64_t*)row)[1] = u_old;
    if (over && !st[SB_STATUS]) { st[SB_STATUS] |= 128ull; st[SB_U_NEW] = 0; st[SB_N_NEW] = 0; }
  }
}

// signature rows of all ranks -> sig_g[ug_old + prefix(rank) + j], owner map; gst[SB_U_NEW] = total, gst[SB_G_BASE] = this rank's first
__global__ __launch_bounds__(256) void ingest_sig_rows_kernel(const uint8_t* __restrict__ rows, uint64_t row_bytes, uint32_t world, uint32_t rank,
                                                               uint64_t sig_cap, uint32_t* __restrict__ sig_g, uint32_t* __restrict__ g_owner,
                                                               uint64_t* __restrict__ g_local, uint64_t max_stored_g, uint64_t* gst, uint64_t* st) {
  const uint64_t ug_old = gst[SB_U_OLD];
  uint64_t total = 0, mine_before = 0;
  bool bad = false;
  for (uint32_t q = 0; q < world; q++) {
    uint64_t c = *(const uint64_t*)(rows + (size_t)q * row_bytes);
    if (c > sig_cap) { bad = true; c = 0; }
    if (q < rank) mine_before += c;
    total += c;
  }
  const bool over = ug_old + total > max_stored_g;
  const bool dead = st[SB_STATUS] != 0;
  if (bad || over || dead) total = 0;
  if (total) {
    uint64_t before = 0;
    for (uint32_t q = 0; q < world; q++) {
      const uint8_t* row = rows + (size_t)q * row_bytes;
      const uint64_t c = ((const uint64_t*)row)[0], lo = ((const uint64_t*)row)[1];
      const uint4* src = (const uint4*)(row + SIG_HDR);
      uint4* dst = (uint4*)(sig_g + 128 * (ug_old + before));
      for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < c * (SIG_BYTES / 16); i += (uint64_t)gridDim.x * 256) dst[i] = src[i];
      for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < c; j += (uint64_t)gridDim.x * 256) { g_owner[ug_old + before + j] = q; g_local[ug_old + before + j] = lo + j; }
      before += c;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    gst[SB_U_NEW] = total;
    gst[SB_G_BASE] = ug_old + mine_before;
    if (!dead && (bad || over)) { st[SB_STATUS] |= bad ? 8ull : 2ull; st[SB_U_NEW] = 0; st[SB_N_NEW] = 0; }
  }
}

// dictionary of every new stored chunk of this rank: own -> its chunk id; remote -> ghost chunk ghost0 + j and request j = (owner, slot),
// requests grouped by owner (counts[q], counts[world] = total).  One workgroup.
__global__ __launch_bounds__(1024) void resolve_bases_kernel(const int64_t* __restrict__ base_g, const uint32_t* __restrict__ keys_g, uint32_t bands,
                                                              const uint32_t* __restrict__ g_owner, const uint64_t* __restrict__ g_local,
                                                              const uint64_t* __restrict__ uniq_all, uint32_t world, uint32_t rank, uint64_t ghost0,
                                                              const uint64_t* gst, const uint64_t* st, uint64_t* __restrict__ ug, int64_t* __restrict__ base_global,
                                                              int64_t* __restrict__ base_all, uint32_t* __restrict__ band_keys, uint64_t* __restrict__ sel_ids,
                                                              int64_t* __restrict__ sel_base, uint64_t* __restrict__ n_sel, uint64_t* __restrict__ req_counts,
                                                              uint64_t* __restrict__ req_slots) {
  __shared__ uint32_t cnt[256], cur[256];
  const uint64_t u_old = st[SB_U_OLD], m0 = gst[SB_G_BASE];
  const uint64_t nu = (st[SB_STATUS] || gst[SB_U_NEW] == 0) ? 0ull : st[SB_U_NEW];
  const uint32_t t = threadIdx.x;
  if (t < 256) { cnt[t] = 0; cur[t] = 0; }
  __syncthreads();
  for (uint64_t k = t; k < nu; k += 1024) {
    const int64_t bg = base_g[m0 + k];
    if (bg >= 0 && g_owner[bg] != rank) atomicAdd(&cnt[g_owner[bg]], 1u);
  }
  __syncthreads();
  if (t == 0) { uint32_t run = 0; for (uint32_t q = 0; q < world; q++) { const uint32_t c = cnt[q]; req_counts[q] = c; cur[q] = run; run += c; } req_counts[world] = run; *n_sel = nu; }
  __syncthreads();
  for (uint64_t k = t; k < nu; k += 1024) {
    const uint64_t gi = m0 + k;
    const int64_t bg = base_g[gi];
    ug[u_old + k] = gi; base_global[u_old + k] = bg;
    for (uint32_t b = 0; b < bands; b++) band_keys[(u_old + k) * bands + b] = keys_g[gi * bands + b];
    sel_ids[k] = uniq_all[u_old + k];
    int64_t bl = -1, bc = -1;
    if (bg >= 0) {
      const uint32_t ow = g_owner[bg];
      const uint64_t loc = g_local[bg];
      if (ow == rank) { bl = (int64_t)loc; bc = (int64_t)uniq_all[loc]; }
      else { const uint32_t j = atomicAdd(&cur[ow], 1u); req_slots[j] = loc; bc = (int64_t)(ghost0 + j); }
    }
    base_all[u_old + k] = bl; sel_base[k] = bc;
  }
}

__global__ void advance_g_kernel(uint64_t* gst, const uint64_t* st) {
  if (st[SB_STATUS]) return;
  gst[SB_U_OLD] += gst[SB_U_NEW];
}

struct Ws {
  uint64_t* cuts_local; uint64_t* n_cuts; uint32_t* l2_status; uint64_t* sel_ids; int64_t* sel_base; uint64_t* n_sel;
  uint64_t* out_off; uint8_t* kind; uint32_t* dfl_status; uint8_t* row;
  uint8_t* l2_ws; size_t l2_bytes; uint8_t* sha_ws; size_t sha_bytes; uint8_t* mh_ws; size_t mh_bytes; uint8_t* dfl_ws; size_t dfl_bytes;
  size_t total;
};

}  // namespace sb

size_t hmse_l2_workspace_bytes_impl(uint64_t n, uint32_t n_seg, const hmse_cfg* cfg);
size_t hmse_l3_sha256_workspace_bytes_impl(uint64_t n_chunks);
size_t hmse_l4_minhash_workspace_bytes_impl(uint64_t n_chunks);
size_t hmse_l1_deflate_workspace_bytes_impl(uint64_t n_chunks, const hmse_cfg* cfg);

static uint32_t sb_n_seg(uint64_t bytes, const hmse_cfg* cfg) { return (uint32_t)((bytes + cfg->seg_size - 1) / cfg->seg_size); }
// worst-case chunk count of a piece of `cap_bytes` (the NOMINAL piece size: it sizes rows, grids and workspaces for every piece)
static uint64_t sb_cap_chunks(uint64_t cap_bytes, const hmse_cfg* cfg) { return cap_bytes / cfg->min_size + sb_n_seg(cap_bytes, cfg) + 2; }

static sb::Ws sb_carve(void* ws, uint64_t cap_bytes, const hmse_cfg* cfg) {
  WsCarver w(ws, ~(size_t)0);
  sb::Ws r;
  const uint64_t cap = sb_cap_chunks(cap_bytes, cfg);
  r.cuts_local = w.take<uint64_t>(cap + 1);
  r.n_cuts = w.take<uint64_t>(1);
  r.l2_status = w.take<uint32_t>(1);
  r.sel_ids = w.take<uint64_t>(cap);
  r.sel_base = w.take<int64_t>(cap);
  r.n_sel = w.take<uint64_t>(1);
  r.out_off = w.take<uint64_t>(cap + 1);
  r.kind = w.take<uint8_t>(cap);
  r.dfl_status = w.take<uint32_t>(1);
  r.row = w.take<uint8_t>(sb::ROW_HDR + 32 * cap);
  r.l2_bytes = hmse_l2_workspace_bytes_impl(cap_bytes, sb_n_seg(cap_bytes, cfg), cfg) + 4096;
  r.l2_ws = w.take<uint8_t>(r.l2_bytes);
  r.sha_bytes = hmse_l3_sha256_workspace_bytes_impl(cap);
  r.sha_ws = w.take<uint8_t>(r.sha_bytes);
  r.mh_bytes = hmse_l4_minhash_workspace_bytes_impl(cap);
  r.mh_ws = w.take<uint8_t>(r.mh_bytes);
  // DEFLATE: fixed part + the record area for the worst case (every chunk with a dictionary: 5 * len + 1317 + padding to 256)
  r.dfl_bytes = hmse_l1_deflate_workspace_bytes_impl(cap, cfg) + (5 * cap_bytes + 1600 * cap) + 4096;
  r.dfl_ws = w.take<uint8_t>(r.dfl_bytes);
  r.total = w.off;
  return r;
}

extern "C" uint64_t hmse_stream_batch_workspace_bytes(uint64_t cap_bytes, const hmse_cfg* cfg) {
  if (hmse_cfg_validate_impl(cfg) != 0 || cap_bytes == 0) return 0;
  return sb_carve(nullptr, cap_bytes, cfg).total;
}

extern "C" int hmse_stream_workspace_init(void* ws, size_t ws_bytes, uint64_t cap_bytes, const hmse_cfg* cfg, void* stream) {
  if (hmse_cfg_validate_impl(cfg) != 0 || cap_bytes == 0 || !ws) return HMSE_EINVAL;
  const sb::Ws w = sb_carve(ws, cap_bytes, cfg);
  if (ws_bytes < w.total) return HMSE_ENOSPC;
  return hmse_l4_minhash_memo_init(w.mh_ws, w.mh_bytes, cfg, (hipStream_t)stream);
}

extern "C" uint64_t hmse_stream_row_bytes(uint64_t cap_bytes, const hmse_cfg* cfg) {
  if (hmse_cfg_validate_impl(cfg) != 0 || cap_bytes == 0) return 0;
  return sb::ROW_HDR + 32 * sb_cap_chunks(cap_bytes, cfg);
}

// ---- phase A ------------------------------------------------------------------------------------------------------------
static int piece_hash(uint8_t* data, uint64_t data_cap, uint64_t piece_bytes, uint64_t cap_bytes, const uint64_t* seg_off, uint32_t n_seg,
                      const hmse_cfg* cfg, uint64_t* state, uint64_t* cuts_all, uint64_t max_chunks, uint8_t* row, const sb::Ws& w,
                      hipStream_t stream) {
  const uint64_t cap = sb_cap_chunks(cap_bytes, cfg);
  if (piece_bytes == 0) {   // this rank has nothing in this batch: N_NEW = 0, an empty row
    sb::append_cuts_kernel<<<dim3(1), dim3(256), 0, stream>>>(nullptr, nullptr, nullptr, cuts_all, state, max_chunks, cap, row);
    HMSE_LAUNCH_CHECK();
    return HMSE_OK;
  }
  int rc;
  // L2 on the piece's bytes (data + state[SB_OFF], read on the device)
  if ((rc = hmse_l2_cdc_impl(data, state + SB_OFF, piece_bytes, seg_off, n_seg, cfg, w.cuts_local, cap + 1, w.n_cuts, w.l2_status, w.l2_ws,
                             w.l2_bytes, stream)) != HMSE_OK) return rc;
  uint32_t ab = (uint32_t)((cap + 255) / 256); if (ab > 1024) ab = 1024;
  sb::append_cuts_kernel<<<dim3(ab), dim3(256), 0, stream>>>(w.cuts_local, w.n_cuts, w.l2_status, cuts_all, state, max_chunks, cap, row);
  // L3: digests of the new chunks go into the exchange row
  if ((rc = hmse_l3_sha256_dyn(data, data_cap, cuts_all, nullptr, row + sb::ROW_HDR, state, cap, w.sha_ws, w.sha_bytes, stream)) != HMSE_OK) return rc;
  HMSE_LAUNCH_CHECK();
  return HMSE_OK;
}

// ---- phase B ------------------------------------------------------------------------------------------------------------
static int piece_encode(uint8_t* data, uint64_t data_cap, uint64_t piece_bytes, uint64_t cap_bytes, const hmse_cfg* cfg, uint64_t* state,
                        const uint8_t* rows, uint32_t world, uint32_t rank, const uint64_t* cuts_all, uint64_t* gidx, uint8_t* digests_g,
                        uint64_t max_chunks_g, uint64_t* first_occ_g, uint32_t* refcount_g, uint32_t* l3_table, uint64_t l3_slots,
                        uint64_t* uniq_all, uint64_t max_unique, uint32_t* sig_all, uint32_t* band_keys, int64_t* base_all, uint32_t* lsh_tables,
                        uint64_t lsh_slots, uint8_t* kind_all, uint64_t* stream_off_all, uint8_t* out, uint64_t out_cap, const sb::Ws& w,
                        hipStream_t stream) {
  const uint64_t cap = sb_cap_chunks(cap_bytes, cfg);
  const uint64_t row_bytes = sb::ROW_HDR + 32 * cap;
  int rc;
  const uint64_t gthreads = 2ull * world * cap;
  sb::ingest_rows_kernel<<<dim3((uint32_t)((gthreads + 255) / 256)), dim3(256), 0, stream>>>(rows, row_bytes, world, rank, cap, digests_g, max_chunks_g,
                                                                                            gidx, state);
  // persistent L3 index over the global chunk order: every rank inserts the same digests under the same indices
  if ((rc = hmse_l3_index_update_dyn(digests_g, first_occ_g, refcount_g, l3_table, l3_slots, state + SB_G_OLD, (uint64_t)world * cap, stream)) != HMSE_OK)
    return rc;
  sb::select_uniq_kernel<<<dim3(1), dim3(1024), 0, stream>>>(first_occ_g, uniq_all, state, max_unique);
  // L4: signatures of the new stored chunks, persistent band tables (this rank's stored chunks: a dictionary must be resident)
  if ((rc = hmse_l4_minhash_dyn(data, data_cap, cuts_all, uniq_all, sig_all, state, cap, cfg, w.mh_ws, w.mh_bytes, stream)) != HMSE_OK) return rc;
  if ((rc = hmse_l4_lsh_update_dyn(sig_all, band_keys, base_all, lsh_tables, lsh_slots, state, cap, cfg, stream)) != HMSE_OK) return rc;
  // L1: dictionary DEFLATE of the new stored chunks (a dictionary may be any stored chunk of this rank's stream so far)
  sb::prep_deflate_kernel<<<dim3((uint32_t)((cap + 255) / 256)), dim3(256), 0, stream>>>(uniq_all, base_all, state, w.sel_ids, w.sel_base, w.n_sel);
  if ((rc = hmse_l1_deflate_dyn(data, data_cap, cuts_all, w.sel_ids, w.sel_base, w.n_sel, cap, state + SB_S_OLD, cfg, out, out_cap, w.out_off, w.kind,
                                w.dfl_status, w.dfl_ws, w.dfl_bytes, stream)) != HMSE_OK) return rc;
  uint32_t ab = (uint32_t)((cap + 255) / 256); if (ab > 1024) ab = 1024;
  sb::commit_kernel<<<dim3(ab), dim3(256), 0, stream>>>(w.kind, w.out_off, cap, kind_all, stream_off_all, w.dfl_status, state);
  sb::advance_kernel<<<dim3(1), dim3(1), 0, stream>>>(state, piece_bytes);
  HMSE_LAUNCH_CHECK();
  return HMSE_OK;
}

static bool sb_piece_args_ok(uint64_t piece_bytes, uint64_t cap_bytes, uint32_t n_seg, const hmse_cfg* cfg) {
  if (hmse_cfg_validate_impl(cfg) != 0) return false;
  if ((cfg->layers & 15u) != 15u) return false;   // the streaming front end runs the full L1-L4 pipeline
  if (cap_bytes == 0 || piece_bytes > cap_bytes) return false;
  if (piece_bytes != 0 && n_seg == 0) return false;
  if (n_seg > sb_n_seg(cap_bytes, cfg)) return false;
  return true;
}

extern "C" int hmse_stream_piece_hash(uint8_t* data, uint64_t data_cap, uint64_t piece_bytes, uint64_t cap_bytes, const uint64_t* seg_off,
                                      uint32_t n_seg, const hmse_cfg* cfg, uint64_t* state, uint64_t* cuts_all, uint64_t max_chunks,
                                      uint8_t* row, void* ws, size_t ws_bytes, void* stream_) {
  if (!sb_piece_args_ok(piece_bytes, cap_bytes, n_seg, cfg)) return HMSE_EINVAL;
  if (!data || !state || !cuts_all || !row || (piece_bytes && !seg_off)) return HMSE_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  (void)hipGetLastError();
  const sb::Ws w = sb_carve(ws, cap_bytes, cfg);
  if (!ws || ws_bytes < w.total) return HMSE_ENOSPC;
  return piece_hash(data, data_cap, piece_bytes, cap_bytes, seg_off, n_seg, cfg, state, cuts_all, max_chunks, row, w, stream);
}

extern "C" int hmse_stream_piece_encode(uint8_t* data, uint64_t data_cap, uint64_t piece_bytes, uint64_t cap_bytes, const hmse_cfg* cfg,
                                        uint64_t* state, const uint8_t* rows, uint32_t world, uint32_t rank, const uint64_t* cuts_all,
                                        uint64_t* gidx, uint8_t* digests_g, uint64_t max_chunks_g, uint64_t* first_occ_g, uint32_t* refcount_g,
                                        uint32_t* l3_table, uint64_t l3_slots, uint64_t* uniq_all, uint64_t max_unique, uint32_t* sig_all,
                                        uint32_t* band_keys, int64_t* base_all, uint32_t* lsh_tables, uint64_t lsh_slots, uint8_t* kind_all,
                                        uint64_t* stream_off_all, uint8_t* out, uint64_t out_cap, void* ws, size_t ws_bytes, void* stream_) {
  if (!sb_piece_args_ok(piece_bytes, cap_bytes, piece_bytes ? 1u : 0u, cfg)) return HMSE_EINVAL;
  if (!data || !state || !rows || !cuts_all || !digests_g || !first_occ_g || !refcount_g || !l3_table || !uniq_all || !sig_all || !band_keys ||
      !base_all || !lsh_tables || !kind_all || !stream_off_all || !out || world == 0 || world > 256 || rank >= world)
    return HMSE_EINVAL;
  if (world > 1 && !gidx) return HMSE_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  (void)hipGetLastError();
  const sb::Ws w = sb_carve(ws, cap_bytes, cfg);
  if (!ws || ws_bytes < w.total) return HMSE_ENOSPC;
  return piece_encode(data, data_cap, piece_bytes, cap_bytes, cfg, state, rows, world, rank, cuts_all, gidx, digests_g, max_chunks_g, first_occ_g,
                      refcount_g, l3_table, l3_slots, uniq_all, max_unique, sig_all, band_keys, base_all, lsh_tables, lsh_slots, kind_all,
                      stream_off_all, out, out_cap, w, stream);
}

// one rank: both phases back to back, the exchange row (kept in the workspace) is the gathered array
extern "C" int hmse_stream_batch(uint8_t* data, uint64_t data_cap, uint64_t batch_bytes, const uint64_t* seg_off, uint32_t n_seg,
                                 const hmse_cfg* cfg, uint64_t* state, uint64_t* cuts_all, uint64_t max_chunks, uint8_t* digests_all,
                                 uint64_t* first_occ, uint32_t* refcount, uint32_t* l3_table, uint64_t l3_slots, uint64_t* uniq_all,
                                 uint64_t max_unique, uint32_t* sig_all, uint32_t* band_keys, int64_t* base_all, uint32_t* lsh_tables,
                                 uint64_t lsh_slots, uint8_t* kind_all, uint64_t* stream_off_all, uint8_t* out, uint64_t out_cap,
                                 void* ws, size_t ws_bytes, void* stream_) {
  if (batch_bytes == 0 || !sb_piece_args_ok(batch_bytes, batch_bytes, n_seg, cfg)) return HMSE_EINVAL;
  if (!data || !seg_off || !state || !cuts_all || !digests_all || !first_occ || !refcount || !l3_table || !uniq_all || !sig_all || !band_keys ||
      !base_all || !lsh_tables || !kind_all || !stream_off_all || !out)
    return HMSE_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  (void)hipGetLastError();
  const sb::Ws w = sb_carve(ws, batch_bytes, cfg);
  if (!ws || ws_bytes < w.total) return HMSE_ENOSPC;
  int rc = piece_hash(data, data_cap, batch_bytes, batch_bytes, seg_off, n_seg, cfg, state, cuts_all, max_chunks, w.row, w, stream);
  if (rc != HMSE_OK) return rc;
  return piece_encode(data, data_cap, batch_bytes, batch_bytes, cfg, state, w.row, 1u, 0u, cuts_all, nullptr, digests_all, max_chunks, first_occ, refcount,
                      l3_table, l3_slots, uniq_all, max_unique, sig_all, band_keys, base_all, lsh_tables, lsh_slots, kind_all, stream_off_all, out,
                      out_cap, w, stream);
}


// ---- global L4 as captured phases (see the kernels above) ----------------------------------------------------------------------
static uint64_t sb_sig_cap(uint64_t cap_bytes, const hmse_cfg* cfg) { return 2 * (cap_bytes / cfg->avg_size) + 64; }   // twice the expected stored chunks of a piece

extern "C" uint64_t hmse_stream_sig_cap(uint64_t cap_bytes, const hmse_cfg* cfg) {
  return (hmse_cfg_validate_impl(cfg) != 0 || cap_bytes == 0) ? 0 : sb_sig_cap(cap_bytes, cfg);
}
extern "C" uint64_t hmse_stream_sig_row_bytes(uint64_t cap_bytes, const hmse_cfg* cfg) {
  return (hmse_cfg_validate_impl(cfg) != 0 || cap_bytes == 0) ? 0 : sb::SIG_HDR + sb::SIG_BYTES * sb_sig_cap(cap_bytes, cfg);
}

static bool gl4_ok(const hmse_gl4* g) {
  return g && g->struct_size == sizeof(hmse_gl4) && g->world >= 1 && g->world <= 256 && g->rank < g->world && g->gstate && g->sig_g && g->band_keys_g &&
         g->base_g && g->lsh_tables_g && g->g_owner && g->g_local && g->ug && g->base_global && g->req_counts && g->req_slots && g->max_stored_g;
}

extern "C" int hmse_stream_piece_sign(uint8_t* data, uint64_t data_cap, uint64_t piece_bytes, uint64_t cap_bytes, const hmse_cfg* cfg, uint64_t* state,
                                      const uint8_t* rows, uint32_t world, uint32_t rank, const uint64_t* cuts_all, uint64_t* gidx, uint8_t* digests_g,
                                      uint64_t max_chunks_g, uint64_t* first_occ_g, uint32_t* refcount_g, uint32_t* l3_table, uint64_t l3_slots,
                                      uint64_t* uniq_all, uint64_t max_unique, uint32_t* sig_all, uint8_t* sig_row, void* ws, size_t ws_bytes, void* stream_) {
  if (!sb_piece_args_ok(piece_bytes, cap_bytes, piece_bytes ? 1u : 0u, cfg)) return HMSE_EINVAL;
  if (!data || !state || !rows || !cuts_all || !digests_g || !first_occ_g || !refcount_g || !l3_table || !uniq_all || !sig_all || !sig_row || world == 0 ||
      world > 256 || rank >= world || (world > 1 && !gidx))
    return HMSE_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  (void)hipGetLastError();
  const sb::Ws w = sb_carve(ws, cap_bytes, cfg);
  if (!ws || ws_bytes < w.total) return HMSE_ENOSPC;
  const uint64_t cap = sb_cap_chunks(cap_bytes, cfg);
  const uint64_t row_bytes = sb::ROW_HDR + 32 * cap;
  int rc;
  const uint64_t gthreads = 2ull * world * cap;
  sb::ingest_rows_kernel<<<dim3((uint32_t)((gthreads + 255) / 256)), dim3(256), 0, stream>>>(rows, row_bytes, world, rank, cap, digests_g, max_chunks_g, gidx, state);
  if ((rc = hmse_l3_index_update_dyn(digests_g, first_occ_g, refcount_g, l3_table, l3_slots, state + SB_G_OLD, (uint64_t)world * cap, stream)) != HMSE_OK) return rc;
  sb::select_uniq_kernel<<<dim3(1), dim3(1024), 0, stream>>>(first_occ_g, uniq_all, state, max_unique);
  if ((rc = hmse_l4_minhash_dyn(data, data_cap, cuts_all, uniq_all, sig_all, state, cap, cfg, w.mh_ws, w.mh_bytes, stream)) != HMSE_OK) return rc;
  sb::sig_row_kernel<<<dim3(1024), dim3(256), 0, stream>>>(sig_all, state, sb_sig_cap(cap_bytes, cfg), sig_row);
  HMSE_LAUNCH_CHECK();
  return HMSE_OK;
}

extern "C" int hmse_stream_piece_bases(uint64_t cap_bytes, const hmse_cfg* cfg, uint64_t* state, const uint8_t* sig_rows, const hmse_gl4* g,
                                       const uint64_t* uniq_all, uint32_t* band_keys, int64_t* base_all, void* ws, size_t ws_bytes, void* stream_) {
  if (hmse_cfg_validate_impl(cfg) != 0 || cap_bytes == 0 || !state || !sig_rows || !gl4_ok(g) || !uniq_all || !band_keys || !base_all) return HMSE_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  (void)hipGetLastError();
  const sb::Ws w = sb_carve(ws, cap_bytes, cfg);
  if (!ws || ws_bytes < w.total) return HMSE_ENOSPC;
  const uint64_t sig_cap = sb_sig_cap(cap_bytes, cfg);
  if (g->sig_cap != sig_cap) return HMSE_EINVAL;
  const uint64_t row_bytes = sb::SIG_HDR + sb::SIG_BYTES * sig_cap;
  sb::ingest_sig_rows_kernel<<<dim3(2048), dim3(256), 0, stream>>>(sig_rows, row_bytes, g->world, g->rank, sig_cap, g->sig_g, g->g_owner, g->g_local,
                                                                 g->max_stored_g, g->gstate, state);
  int rc;
  if ((rc = hmse_l4_lsh_update_dyn(g->sig_g, g->band_keys_g, g->base_g, g->lsh_tables_g, g->lsh_slots_g, g->gstate, (uint64_t)g->world * sig_cap, cfg, stream)) != HMSE_OK)
    return rc;
  sb::resolve_bases_kernel<<<dim3(1), dim3(1024), 0, stream>>>(g->base_g, g->band_keys_g, cfg->bands, g->g_owner, g->g_local, uniq_all, g->world, g->rank,
                                                              g->ghost_chunk0, g->gstate, state, g->ug, g->base_global, base_all, band_keys, w.sel_ids, w.sel_base,
                                                              w.n_sel, g->req_counts, g->req_slots);
  HMSE_LAUNCH_CHECK();
  return HMSE_OK;
}

extern "C" int hmse_stream_piece_encode_g(uint8_t* data, uint64_t data_cap, uint64_t piece_bytes, uint64_t cap_bytes, const hmse_cfg* cfg, uint64_t* state,
                                          uint64_t* gstate, const uint64_t* cuts_all, uint8_t* kind_all, uint64_t* stream_off_all, uint8_t* out,
                                          uint64_t out_cap, void* ws, size_t ws_bytes, void* stream_) {
  if (!sb_piece_args_ok(piece_bytes, cap_bytes, piece_bytes ? 1u : 0u, cfg)) return HMSE_EINVAL;
  if (!data || !state || !gstate || !cuts_all || !kind_all || !stream_off_all || !out) return HMSE_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  (void)hipGetLastError();
  const sb::Ws w = sb_carve(ws, cap_bytes, cfg);
  if (!ws || ws_bytes < w.total) return HMSE_ENOSPC;
  const uint64_t cap = sb_cap_chunks(cap_bytes, cfg);
  int rc;
  if ((rc = hmse_l1_deflate_dyn(data, data_cap, cuts_all, w.sel_ids, w.sel_base, w.n_sel, cap, state + SB_S_OLD, cfg, out, out_cap, w.out_off, w.kind,
                                w.dfl_status, w.dfl_ws, w.dfl_bytes, stream)) != HMSE_OK) return rc;
  uint32_t ab = (uint32_t)((cap + 255) / 256); if (ab > 1024) ab = 1024;
  sb::commit_kernel<<<dim3(ab), dim3(256), 0, stream>>>(w.kind, w.out_off, cap, kind_all, stream_off_all, w.dfl_status, state);
  sb::advance_g_kernel<<<dim3(1), dim3(1), 0, stream>>>(gstate, state);
  sb::advance_kernel<<<dim3(1), dim3(1), 0, stream>>>(state, piece_bytes);
  HMSE_LAUNCH_CHECK();
  return HMSE_OK;
}
