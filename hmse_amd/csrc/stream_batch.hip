// stream_batch.hip — one batch of the streaming front end as ONE enqueue without a host read (BASELINE.json
// configs[4] "hipGraph-captured per-batch pipeline"; the reference's batch loop README.md:1519-1580).
//
// hmse_amd/stream.py's eager path reads device counts on the host between stages (chunk count, stored-chunk count, stream
// bytes) to size the next launch.  Here every stage takes its ranges from a small state block in HBM (common.h, SB_*), grids
// and workspaces are sized for the worst case of the batch (n / min_size chunks), and the chain ends by advancing the state
// itself (streams are appended to `out` behind the earlier batches' at state[SB_S_OLD]) — so the whole sequence L2 -> L3 (hash + persistent index) -> stored-chunk selection -> L4 (MinHash + persistent band
// tables) -> L1 (dictionary DEFLATE) -> index tails can be captured into a hipGraph once and replayed for every batch of that
// size; the host reads the state block when it wants the counts (at the end of the stream, or per batch to fetch the streams).
// Results equal the eager path's bit for bit (tests/test_gpu_stream.py).
#include "common.h"

namespace sb {

// cuts of the batch (batch-local, from L2) -> global cut array; n_new
__global__ __launch_bounds__(256) void append_cuts_kernel(const uint64_t* __restrict__ local, const uint64_t* __restrict__ n_cuts,
                                                           uint64_t* __restrict__ cuts_all, uint64_t* st, uint64_t max_chunks) {
  const uint64_t n_old = st[SB_N_OLD], off = st[SB_OFF];
  uint64_t n_new = *n_cuts;
  const bool fits = n_old + n_new <= max_chunks;
  if (!fits) n_new = 0;   // capacity exceeded: the batch is dropped and flagged, nothing is written out of bounds
  for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < n_new; j += (uint64_t)gridDim.x * 256) cuts_all[n_old + 1 + j] = local[j + 1] + off;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st[SB_N_NEW] = n_new;
    if (!fits) st[SB_STATUS] |= 1ull;
  }
}

// stored chunks of the batch = chunks that are their own first occurrence: ascending append to uniq_all, count -> state.
// One workgroup: the batch has at most a few hundred thousand chunks.
__global__ __launch_bounds__(1024) void select_uniq_kernel(const uint64_t* __restrict__ first_occ, uint64_t* __restrict__ uniq_all,
                                                            uint64_t* st, uint64_t max_unique) {
  __shared__ uint32_t red[1024 / 64 + 1];
  __shared__ uint64_t run;
  const uint64_t n_old = st[SB_N_OLD], n_new = st[SB_N_NEW], u_old = st[SB_U_OLD];
  if (threadIdx.x == 0) run = 0;
  __syncthreads();
  for (uint64_t b0 = 0; b0 < n_new; b0 += 1024) {
    const uint64_t i = n_old + b0 + threadIdx.x;
    const bool mine = b0 + threadIdx.x < n_new && first_occ[i] == i;
    uint32_t total;
    const uint32_t ex = block_exclusive_scan<1024>(mine ? 1u : 0u, red, &total);
    const uint64_t r = run;
    if (mine && u_old + r + ex < max_unique) uniq_all[u_old + r + ex] = i;
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

// the batch's DEFLATE selection as contiguous arrays: chunk ids of its stored chunks and their dictionaries as chunk ids
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

// tails of kind / stream offsets, then the state moves on to the next batch
__global__ __launch_bounds__(256) void commit_kernel(const uint8_t* __restrict__ kind_b, const uint64_t* __restrict__ out_off_b, uint64_t cap_sel,
                                                      uint8_t* __restrict__ kind_all, uint64_t* __restrict__ stream_off_all,
                                                      const uint32_t* __restrict__ dfl_status, const uint32_t* __restrict__ l2_status, uint64_t* st) {
  const uint64_t u_old = st[SB_U_OLD], nu = st[SB_U_NEW], s_old = st[SB_S_OLD];
  const uint64_t total = nu ? out_off_b[cap_sel] : 0ull;
  for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k < nu; k += (uint64_t)gridDim.x * 256) {
    kind_all[u_old + k] = kind_b[k];
    stream_off_all[u_old + k + 1] = (k + 1 < nu ? out_off_b[k + 1] : total) + s_old;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st[SB_S_NEW] = total;
    if (*dfl_status) st[SB_STATUS] |= (uint64_t)*dfl_status << 8;
    if (*l2_status) st[SB_STATUS] |= 4ull;
  }
}
__global__ void advance_kernel(uint64_t* st, uint64_t batch_bytes) {
  st[SB_OFF] += batch_bytes;
  st[SB_N_OLD] += st[SB_N_NEW];
  st[SB_U_OLD] += st[SB_U_NEW];
  st[SB_S_OLD] += st[SB_S_NEW];
}

struct Ws {
  uint64_t* cuts_local; uint64_t* n_cuts; uint32_t* l2_status; uint64_t* sel_ids; int64_t* sel_base; uint64_t* n_sel;
  uint64_t* out_off; uint8_t* kind; uint32_t* dfl_status;
  uint8_t* l2_ws; size_t l2_bytes; uint8_t* sha_ws; size_t sha_bytes; uint8_t* mh_ws; size_t mh_bytes; uint8_t* dfl_ws; size_t dfl_bytes;
  size_t total;
};

}  // namespace sb

size_t hmse_l2_workspace_bytes_impl(uint64_t n, uint32_t n_seg, const hmse_cfg* cfg);
size_t hmse_l3_sha256_workspace_bytes_impl(uint64_t n_chunks);
size_t hmse_l4_minhash_workspace_bytes_impl(uint64_t n_chunks);
size_t hmse_l1_deflate_workspace_bytes_impl(uint64_t n_chunks, const hmse_cfg* cfg);

static uint64_t sb_cap_chunks(uint64_t batch_bytes, uint32_t n_seg, const hmse_cfg* cfg) { return batch_bytes / cfg->min_size + n_seg + 2; }

static sb::Ws sb_carve(void* ws, uint64_t batch_bytes, uint32_t n_seg, const hmse_cfg* cfg) {
  WsCarver w(ws, ~(size_t)0);
  sb::Ws r;
  const uint64_t cap = sb_cap_chunks(batch_bytes, n_seg, cfg);
  r.cuts_local = w.take<uint64_t>(cap + 1);
  r.n_cuts = w.take<uint64_t>(1);
  r.l2_status = w.take<uint32_t>(1);
  r.sel_ids = w.take<uint64_t>(cap);
  r.sel_base = w.take<int64_t>(cap);
  r.n_sel = w.take<uint64_t>(1);
  r.out_off = w.take<uint64_t>(cap + 1);
  r.kind = w.take<uint8_t>(cap);
  r.dfl_status = w.take<uint32_t>(1);
  r.l2_bytes = hmse_l2_workspace_bytes_impl(batch_bytes, n_seg, cfg) + 4096;
  r.l2_ws = w.take<uint8_t>(r.l2_bytes);
  r.sha_bytes = hmse_l3_sha256_workspace_bytes_impl(cap);
  r.sha_ws = w.take<uint8_t>(r.sha_bytes);
  r.mh_bytes = hmse_l4_minhash_workspace_bytes_impl(cap);
  r.mh_ws = w.take<uint8_t>(r.mh_bytes);
  // DEFLATE: fixed part + the record area for the worst case (every chunk stored, every one with a dictionary)
  r.dfl_bytes = hmse_l1_deflate_workspace_bytes_impl(cap, cfg) + 2 * (5 * batch_bytes + 1600 * cap) + 4096;
  r.dfl_ws = w.take<uint8_t>(r.dfl_bytes);
  r.total = w.off;
  return r;
}

extern "C" uint64_t hmse_stream_batch_workspace_bytes(uint64_t batch_bytes, const hmse_cfg* cfg) {
  if (hmse_cfg_validate_impl(cfg) != 0 || batch_bytes == 0) return 0;
  const uint32_t n_seg = (uint32_t)((batch_bytes + cfg->seg_size - 1) / cfg->seg_size);
  return sb_carve(nullptr, batch_bytes, n_seg, cfg).total;
}

extern "C" int hmse_stream_batch(uint8_t* data, uint64_t data_cap, uint64_t batch_bytes, const uint64_t* seg_off, uint32_t n_seg,
                                 const hmse_cfg* cfg, uint64_t* state, uint64_t* cuts_all, uint64_t max_chunks, uint8_t* digests_all,
                                 uint64_t* first_occ, uint32_t* refcount, uint32_t* l3_table, uint64_t l3_slots, uint64_t* uniq_all,
                                 uint64_t max_unique, uint32_t* sig_all, uint32_t* band_keys, int64_t* base_all, uint32_t* lsh_tables,
                                 uint64_t lsh_slots, uint8_t* kind_all, uint64_t* stream_off_all, uint8_t* out, uint64_t out_cap,
                                 void* ws, size_t ws_bytes, void* stream_) {
  if (hmse_cfg_validate_impl(cfg) != 0) return HMSE_EINVAL;
  if (!data || !seg_off || !state || !cuts_all || !digests_all || !first_occ || !refcount || !l3_table || !uniq_all || !sig_all || !band_keys ||
      !base_all || !lsh_tables || !kind_all || !stream_off_all || !out || batch_bytes == 0 || n_seg == 0)
    return HMSE_EINVAL;
  if ((cfg->layers & 15u) != 15u) return HMSE_EINVAL;   // the streaming front end runs the full L1-L4 pipeline
  hipStream_t stream = (hipStream_t)stream_;
  (void)hipGetLastError();
  sb::Ws w = sb_carve(ws, batch_bytes, n_seg, cfg);
  if (!ws || ws_bytes < w.total) return HMSE_ENOSPC;
  const uint64_t cap = sb_cap_chunks(batch_bytes, n_seg, cfg);
  int rc;
  // L2 on the batch's bytes (data + state[SB_OFF], read on the device)
  if ((rc = hmse_l2_cdc_impl(data, state + SB_OFF, batch_bytes, seg_off, n_seg, cfg, w.cuts_local, cap + 1, w.n_cuts, w.l2_status, w.l2_ws,
                             w.l2_bytes, stream)) != HMSE_OK) return rc;
  uint32_t ab = (uint32_t)((cap + 255) / 256); if (ab > 1024) ab = 1024;
  sb::append_cuts_kernel<<<dim3(ab), dim3(256), 0, stream>>>(w.cuts_local, w.n_cuts, cuts_all, state, max_chunks);
  // L3: digests of the new chunks, persistent index
  if ((rc = hmse_l3_sha256_dyn(data, data_cap, cuts_all, digests_all, state, cap, w.sha_ws, w.sha_bytes, stream)) != HMSE_OK) return rc;
  if ((rc = hmse_l3_index_update_dyn(digests_all, first_occ, refcount, l3_table, l3_slots, state, cap, stream)) != HMSE_OK) return rc;
  sb::select_uniq_kernel<<<dim3(1), dim3(1024), 0, stream>>>(first_occ, uniq_all, state, max_unique);
  // L4: signatures of the new stored chunks, persistent band tables
  if ((rc = hmse_l4_minhash_dyn(data, data_cap, cuts_all, uniq_all, sig_all, state, cap, cfg, w.mh_ws, w.mh_bytes, stream)) != HMSE_OK) return rc;
  if ((rc = hmse_l4_lsh_update_dyn(sig_all, band_keys, base_all, lsh_tables, lsh_slots, state, cap, cfg, stream)) != HMSE_OK) return rc;
  // L1: dictionary DEFLATE of the new stored chunks (a dictionary may be any stored chunk of the stream so far)
  sb::prep_deflate_kernel<<<dim3((uint32_t)((cap + 255) / 256)), dim3(256), 0, stream>>>(uniq_all, base_all, state, w.sel_ids, w.sel_base, w.n_sel);
  if ((rc = hmse_l1_deflate_dyn(data, data_cap, cuts_all, w.sel_ids, w.sel_base, w.n_sel, cap, state + SB_S_OLD, cfg, out, out_cap, w.out_off, w.kind,
                                w.dfl_status, w.dfl_ws, w.dfl_bytes, stream)) != HMSE_OK) return rc;
  sb::commit_kernel<<<dim3(ab), dim3(256), 0, stream>>>(w.kind, w.out_off, cap, kind_all, stream_off_all, w.dfl_status, w.l2_status, state);
  sb::advance_kernel<<<dim3(1), dim3(1), 0, stream>>>(state, batch_bytes);
  HMSE_LAUNCH_CHECK();
  return HMSE_OK;
}
