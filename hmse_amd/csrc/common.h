// common.h — shared device/host helpers for the gfx950 HMSE kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "../../include/hmse.h"

#define HMSE_LAUNCH_CHECK()                                 \
  do {                                                      \
    hipError_t e__ = hipGetLastError();                     \
    if (e__ != hipSuccess) return HMSE_EHIP;                \
  } while (0)

#define HMSE_HIP(x)                                         \
  do {                                                      \
    if ((x) != hipSuccess) return HMSE_EHIP;                \
  } while (0)

static inline size_t hmse_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// carve a sub-buffer out of the caller's workspace (256-B aligned pieces)
struct WsCarver {
  uint8_t* base;
  size_t cap, off;
  __host__ WsCarver(void* p, size_t c) : base((uint8_t*)p), cap(c), off(0) {}
  template <typename T>
  __host__ T* take(size_t count) {
    size_t bytes = hmse_align_up(count * sizeof(T), 256);
    T* r = (T*)(base ? base + off : nullptr);
    off += bytes;
    return r;
  }
  __host__ bool ok() const { return base != nullptr && off <= cap; }
};

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }
// number of set bits of m below this lane (v_mbcnt: no lane-mask registers)
__device__ __forceinline__ uint32_t mbcnt64(uint64_t m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

__device__ __forceinline__ uint4 load_u4_unaligned(const uint8_t* p) {
  uint4 v;
  __builtin_memcpy(&v, p, 16);
  return v;
}
__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t* p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);
  return v;
}
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return __builtin_rotateleft32(x, r); }
__device__ __forceinline__ uint32_t rotr32(uint32_t x, int r) { return __builtin_rotateright32(x, r); }

// inclusive prefix sum over the 64 lanes on the DPP path of the vector ALU (row shifts inside the 16-lane rows, then the two row
// broadcasts of the GFX9 family): no LDS crossbar round trips (ds_bpermute, which is what __shfl_up compiles to)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);   // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);   // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);   // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);   // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1 and 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2 and 3
  return v;
}

// the maximum over the 64 lanes, in every lane (same DPP ladder; 0 is the identity)
__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
#define HMSE_DPP_MAX(ctrl, rows) { const uint32_t o__ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rows, 0xF, false); v = v > o__ ? v : o__; }
  HMSE_DPP_MAX(0x111, 0xF) HMSE_DPP_MAX(0x112, 0xF) HMSE_DPP_MAX(0x114, 0xF) HMSE_DPP_MAX(0x118, 0xF) HMSE_DPP_MAX(0x142, 0xA) HMSE_DPP_MAX(0x143, 0xC)
#undef HMSE_DPP_MAX
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(v), 63); }

// workgroup exclusive scan of one u32 per thread (NT threads, NT <= 1024). `red` = LDS u32[NT/64 + 1].
template <int NT>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* red, uint32_t* total) {
  const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
  const uint32_t inc = wave_incl_scan(v);
  if (lane == 63) red[wave] = inc;
  __syncthreads();
  uint32_t wbase = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; w++) {
    uint32_t c = red[w];
    if ((uint32_t)w < wave) wbase += c;
    tot += c;
  }
  __syncthreads();
  *total = tot;
  return wbase + inc - v;
}

// Every byte this library clears is cleared by a KERNEL (hmse_fill_async), never by hipMemsetAsync / hipMemcpyAsync: a captured
// chain then consists of kernel nodes only.  Reason (round 3, tools/graph_two_execs_probe.py, profiles/r3/r3_hipgraph_packet_capture.txt):
// with ROCm 7.0's default DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 a hipGraph of the phase-A chain — kernels interleaved with small
// memset nodes and one 4-byte device-to-device memcpy node — ran correctly on its FIRST launch and with garbage kernel
// arguments from its SECOND launch on (the runtime replays AQL packets it recorded during the first launch).
// `bytes` and the address must be multiples of 4.
int hmse_fill_async(void* p, uint32_t byte_value, size_t bytes, hipStream_t stream);
#define HMSE_FILL(p, byte_value, bytes, stream)                                        \
  do {                                                                                 \
    int rc__ = hmse_fill_async((p), (byte_value), (bytes), (stream));                  \
    if (rc__ != HMSE_OK) return rc__;                                                  \
  } while (0)

// diagnostics: event pair around a stage's dominant kernel (see hmse_profile_enable)
extern int g_hmse_prof;
extern unsigned long long* g_hmse_prof_ctr;   // DEVICE u64[32] work counters per profile slot (nullptr until profiling was enabled once)
void hmse_prof_begin(int stage, hipStream_t s);
void hmse_prof_end(int stage, hipStream_t s);
#define PROF_BEGIN(stage, s) do { if (g_hmse_prof) hmse_prof_begin(stage, s); } while (0)
#define PROF_END(stage, s) do { if (g_hmse_prof) hmse_prof_end(stage, s); } while (0)

// internal cross-file declarations
int hmse_cfg_validate_impl(const hmse_cfg* cfg);
void hmse_cdc_masks_hi(const hmse_cfg* cfg, uint32_t* ms_hi, uint32_t* ml_hi);
uint32_t hmse_deflate_depth(const hmse_cfg* cfg);

// ---- device-side state of the captured per-batch chain (hmse_stream_batch, stream_batch.hip) ---------------------
// u64 words in HBM, read by every stage instead of host-side counts, so that the whole chain of one batch can be
// enqueued (and captured into a hipGraph) without a single host read.
enum {
  SB_OFF = 0,        // byte offset of this batch in the resident corpus (= bytes ingested before it)
  SB_N_OLD = 1,      // chunks before this batch
  SB_N_NEW = 2,      // chunks of this batch            (written by the chain)
  SB_U_OLD = 3,      // stored chunks before this batch
  SB_U_NEW = 4,      // stored chunks of this batch     (written by the chain)
  SB_S_OLD = 5,      // stream bytes before this batch
  SB_S_NEW = 6,      // stream bytes of this batch      (written by the chain)
  SB_STATUS = 7,     // sticky error bits: once non-zero every later batch of the stream is a no-op
  // multi-rank streaming (global chunk order = batch, rank, local index); with one rank these mirror SB_N_OLD / SB_N_NEW
  SB_G_OLD = 8,      // chunks of ALL ranks before this batch
  SB_G_NEW = 9,      // chunks of all ranks in this batch   (written by the chain)
  SB_G_BASE = 10,    // global index of this rank's first chunk of this batch (written by the chain)
  SB_WORDS = 16
};
int hmse_l2_cdc_impl(const uint8_t* data, const uint64_t* data_off_dev, uint64_t n, const uint64_t* seg_off, uint32_t n_seg,
                     const hmse_cfg* cfg, uint64_t* cuts, uint64_t cuts_cap, uint64_t* n_cuts, uint32_t* status, void* ws,
                     size_t ws_bytes, hipStream_t stream);
// digests_batch != nullptr: the digest of the batch's j-th chunk goes to digests_batch + 32 j (the rank's exchange record) instead of
// digests_all + 32 (st[SB_N_OLD] + j)
int hmse_l3_sha256_dyn(const uint8_t* data, uint64_t n_cap, const uint64_t* cuts_all, uint8_t* digests_all, uint8_t* digests_batch,
                       const uint64_t* st, uint64_t cap_chunks, void* ws, size_t ws_bytes, hipStream_t stream);
// rng: DEVICE u64[2] = {first index, count} of the digests that join the table (state + SB_N_OLD, or state + SB_G_OLD)
int hmse_l3_index_update_dyn(const uint8_t* digests_all, uint64_t* first_occ, uint32_t* refcount, uint32_t* table, uint64_t slots,
                             const uint64_t* rng, uint64_t cap_chunks, hipStream_t stream);
int hmse_l4_minhash_memo_init(void* ws, size_t ws_bytes, const hmse_cfg* cfg, hipStream_t stream);
int hmse_l4_minhash_dyn(const uint8_t* data, uint64_t n_cap, const uint64_t* cuts_all, const uint64_t* uniq_all, uint32_t* sig_all,
                        const uint64_t* st, uint64_t cap_chunks, const hmse_cfg* cfg, void* ws, size_t ws_bytes, hipStream_t stream);
int hmse_l4_lsh_update_dyn(const uint32_t* sig_all, uint32_t* band_keys, int64_t* base_all, uint32_t* tables, uint64_t slots,
                           const uint64_t* st, uint64_t cap_chunks, const hmse_cfg* cfg, hipStream_t stream);
int hmse_l1_deflate_dyn(const uint8_t* data, uint64_t n_cap, const uint64_t* cuts_all, const uint64_t* sel_ids, const int64_t* sel_base,
                        const uint64_t* n_sel_dev, uint64_t cap_sel, const uint64_t* out_base_dev, const hmse_cfg* cfg, uint8_t* out,
                        uint64_t out_cap, uint64_t* out_off, uint8_t* kind, uint32_t* status, void* ws, size_t ws_bytes, hipStream_t stream);
