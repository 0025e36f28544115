// l4_minhash.hip — L4a MinHash signatures for gfx950.
//
// Replaces minhash_compute() (README.md:2578-2598):
//   sig[h] = min over 4-byte shingles x of MurmurHash3_x86_32(x, 4, seed_base + h),  h < 128.
// This is the VALU wall of the pipeline (128 murmur finalisers per byte position), so the kernel
// applies the exactness-preserving reductions of SURVEY.md §7:
//   * the seed-independent k = rotl(x*c1,15)*c2 is computed once per shingle;
//   * a minimum over a multiset is the minimum over its SET: each workgroup first de-duplicates
//     the chunk's shingles in an LDS hash set (k is a bijection of x, so the set holds k);
//   * lanes are the 128 seeds (two per lane), so no cross-lane reduction is needed per shingle.
// One chunk per workgroup (4 wavefronts): all four waves fill the set, each wave then owns a
// quarter of the table, compacts it in place with ballot/popcount and streams it (LDS broadcast
// reads) through its 128 running minima; the four partial signatures are min-combined in LDS.
#include "common.h"

constexpr int MH_NT = 256;
constexpr int MH_TBITS = 14;
constexpr int MH_SLOTS = 1 << MH_TBITS;   // 16384 x 4 B = 64 KiB
constexpr int MH_SUB = MH_SLOTS * 3 / 4;  // shingles per pass: load factor <= 0.75 even if all are distinct (text: ~0.4);
                                          // 12288 covers a typical 8-12 KiB chunk in ONE pass, so its whole shingle set is de-duplicated
constexpr uint32_t MH_EMPTY = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t murmur_k(uint32_t x) {
  uint32_t k = x * 0xcc9e2d51u;
  k = rotl32(k, 15);
  return k * 0x1b873593u;
}
// remainder of MurmurHash3_x86_32 for a single 4-byte block: h1 = seed ^ k ... fmix32
__device__ __forceinline__ uint32_t murmur_tail(uint32_t k, uint32_t seed) {
  uint32_t h = seed ^ k;
  h = rotl32(h, 13);
  h = h * 5u + 0xe6546b64u;
  h ^= 4u;
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}

__global__ __launch_bounds__(MH_NT) void l4_minhash_kernel(const uint8_t* __restrict__ data, uint64_t n,
                                                            const uint64_t* __restrict__ cuts,
                                                            const uint64_t* __restrict__ chunk_ids, uint64_t n_sel,
                                                            uint32_t seed_base, uint32_t* __restrict__ sig) {
  __shared__ __attribute__((aligned(16))) uint32_t s_tab[MH_SLOTS];
  __shared__ uint32_t s_sig[MH_NT / 64][128];
  __shared__ uint32_t s_flag;
  const uint64_t sel = blockIdx.x;
  if (sel >= n_sel) return;
  const uint32_t t = threadIdx.x, lane = lane_id(), w = t >> 6;
  const uint64_t c = chunk_ids ? chunk_ids[sel] : sel;
  const uint64_t start = cuts[c];
  const uint64_t len = cuts[c + 1] - start;
  const uint64_t nsh = len >= 4 ? len - 3 : 0;
  const uint32_t seed0 = seed_base + lane, seed1 = seed_base + lane + 64;
  uint32_t m0 = 0xFFFFFFFFu, m1 = 0xFFFFFFFFu;
  constexpr int QUARTER = MH_SLOTS / (MH_NT / 64);
  uint32_t* q = s_tab + w * QUARTER;

  for (uint64_t sub0 = 0; sub0 < nsh; sub0 += MH_SUB) {
    const uint32_t cnt = (uint32_t)((nsh - sub0) < (uint64_t)MH_SUB ? (nsh - sub0) : (uint64_t)MH_SUB);
    // clear the set
    for (uint32_t i = t; i < MH_SLOTS / 4; i += MH_NT) ((uint4*)s_tab)[i] = make_uint4(MH_EMPTY, MH_EMPTY, MH_EMPTY, MH_EMPTY);
    if (t == 0) s_flag = 0;
    __syncthreads();
    // insert k of every shingle of this pass
    const uint8_t* src = data + start + sub0;
    for (uint32_t p = t; p < cnt; p += MH_NT) {
      const uint32_t k = murmur_k(load_u32_unaligned(src + p));
      if (k == MH_EMPTY) {
        s_flag = 1;
      } else {
        uint32_t slot = k >> (32 - MH_TBITS);
        for (;;) {
          const uint32_t old = atomicCAS(&s_tab[slot], MH_EMPTY, k);
          if (old == MH_EMPTY || old == k) break;
          slot = (slot + 1) & (MH_SLOTS - 1);
        }
      }
    }
    __syncthreads();
    // in-place compaction of this wave's quarter (write index never passes the read index)
    uint32_t wr = 0;
    for (uint32_t i = 0; i < (uint32_t)QUARTER; i += 64) {
      const uint32_t v = q[i + lane];
      const bool valid = v != MH_EMPTY;
      const uint64_t m = __ballot(valid);
      if (valid) q[wr + (uint32_t)__builtin_popcountll(m & lanemask_lt())] = v;
      wr += (uint32_t)__builtin_popcountll(m);
    }
    // stream the distinct k values through the 2 seeds of this lane
    const uint32_t wr4 = wr & ~3u;
    for (uint32_t i = 0; i < wr4; i += 4) {
      const uint4 kv = *(const uint4*)(q + i);
      const uint32_t a0 = murmur_tail(kv.x, seed0), a1 = murmur_tail(kv.x, seed1);
      const uint32_t b0 = murmur_tail(kv.y, seed0), b1 = murmur_tail(kv.y, seed1);
      const uint32_t c0 = murmur_tail(kv.z, seed0), c1 = murmur_tail(kv.z, seed1);
      const uint32_t d0 = murmur_tail(kv.w, seed0), d1 = murmur_tail(kv.w, seed1);
      m0 = min(min(m0, min(a0, b0)), min(c0, d0));
      m1 = min(min(m1, min(a1, b1)), min(c1, d1));
    }
    for (uint32_t i = wr4; i < wr; i++) {
      const uint32_t k = q[i];
      m0 = min(m0, murmur_tail(k, seed0));
      m1 = min(m1, murmur_tail(k, seed1));
    }
    if (w == 0 && s_flag) {  // the one k value that collides with the empty marker
      m0 = min(m0, murmur_tail(MH_EMPTY, seed0));
      m1 = min(m1, murmur_tail(MH_EMPTY, seed1));
    }
    __syncthreads();
  }
  s_sig[w][lane] = m0;
  s_sig[w][lane + 64] = m1;
  __syncthreads();
  if (t < 128) {
    uint32_t v = s_sig[0][t];
#pragma unroll
    for (int i = 1; i < MH_NT / 64; i++) v = min(v, s_sig[i][t]);
    sig[sel * 128 + t] = v;
  }
}

size_t hmse_l4_minhash_workspace_bytes_impl(uint64_t) { return 256; }

extern "C" int hmse_l4_minhash(const uint8_t* data, uint64_t n, const uint64_t* cuts, const uint64_t* chunk_ids,
                               uint64_t n_sel, const hmse_cfg* cfg, uint32_t* sig, void* ws, size_t ws_bytes, void* stream_) {
  (void)ws; (void)ws_bytes;
  if (hmse_cfg_validate_impl(cfg) != 0) return HMSE_EINVAL;
  if (n_sel == 0) return HMSE_OK;
  if (!data || !cuts || !sig) return HMSE_EINVAL;
  if (n_sel > 0x7FFFFFFFull) return HMSE_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  (void)hipGetLastError();  // drop stale errors of earlier runtime calls made by the host process
  PROF_BEGIN(HMSE_STAGE_L4_MINHASH, stream);
  l4_minhash_kernel<<<dim3((uint32_t)n_sel), dim3(MH_NT), 0, stream>>>(data, n, cuts, chunk_ids, n_sel, cfg->seed_base, sig);
  PROF_END(HMSE_STAGE_L4_MINHASH, stream);
  HMSE_LAUNCH_CHECK();
  return HMSE_OK;
}
