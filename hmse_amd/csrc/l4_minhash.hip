// l4_minhash.hip — L4a MinHash signatures for gfx950.
//
// Replaces minhash_compute() (README.md:2578-2598):
//   sig[h] = min over 4-byte shingles x of MurmurHash3_x86_32(x, 4, seed_base + h),  h < 128.
// This is the VALU wall of the pipeline (128 murmur finalisers per byte position), so the kernel
// applies the exactness-preserving reductions of SURVEY.md §7:
//   * the seed-independent part R = rotl(rotl(x*c1,15)*c2, 13) is computed once per shingle;
//   * a minimum over a multiset is the minimum over its SET: each workgroup first de-duplicates
//     the chunk's shingles in an LDS hash set (R is a bijection of x, so the set holds R);
//   * lanes are the 128 seeds (two per lane), so no cross-lane reduction is needed per shingle.
// One chunk per workgroup (4 wavefronts): all four waves fill the set, each wave then owns a
// quarter of the table, compacts it in place with ballot/popcount and streams it (LDS broadcast
// reads) through its 128 running minima; the four partial signatures are min-combined in LDS.
#include "common.h"
#include <stdlib.h>

constexpr int MH_DEFAULT_VARIANT = 5;

constexpr int MH_TBITS = 14;
constexpr int MH_SLOTS = 1 << MH_TBITS;   // 16384 x 4 B = 64 KiB
constexpr int MH_SUB = MH_SLOTS * 3 / 4;  // shingles per pass: load factor <= 0.75 even if all are distinct (text: ~0.4);
                                          // 12288 covers a typical 8-12 KiB chunk in ONE pass, so its whole shingle set is de-duplicated
constexpr uint32_t MH_EMPTY = 0xFFFFFFFFu;

// MurmurHash3_x86_32 of one 4-byte block x with seed s:  k = rotl(x*c1,15)*c2;  h = rotl(s ^ k, 13);  h = h*5 + c;
// h ^= 4 (the length);  fmix32.  rotl distributes over xor, so h = rotl(s,13) ^ R with R = rotl(k,13): the set holds
// R (a bijection of x, computed once per position) and every lane keeps rotl(seed,13) of its seeds.
__device__ __forceinline__ uint32_t murmur_R(uint32_t x) {
  uint32_t k = x * 0xcc9e2d51u;
  k = rotl32(k, 15);
  return rotl32(k * 0x1b873593u, 13);
}
// the per-(shingle, seed) remainder.  V = 0: plain C (the compiler's choice of instructions);  V = 1: the first
// xor-shift and the length xor as one three-input v_bitop3 (measured: tools/ubench/valu_rates.hip)
template <int V>
__device__ __forceinline__ uint32_t murmur_tail(uint32_t R, uint32_t sr) {
  uint32_t h = (sr ^ R) * 5u + 0xe6546b64u;
  if (V == 1) {
    const uint32_t hs = h >> 16;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(h) : "v"(h), "v"(hs), "v"(4u));  // h ^ (h >> 16) ^ 4
  } else {
    h ^= 4u;
    h ^= h >> 16;
  }
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}

__device__ __forceinline__ uint32_t min3u(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r;
  asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

template <int NT, int V>
__global__ __launch_bounds__(NT) void l4_minhash_kernel(const uint8_t* __restrict__ data, uint64_t n,
                                                        const uint64_t* __restrict__ cuts,
                                                        const uint64_t* __restrict__ chunk_ids, uint64_t n_sel,
                                                        uint32_t seed_base, uint32_t* __restrict__ sig, const uint64_t* __restrict__ st) {
  __shared__ __attribute__((aligned(16))) uint32_t s_tab[MH_SLOTS];
  __shared__ uint32_t s_sig[NT / 64][128];
  __shared__ uint32_t s_flag;
  if (st) { chunk_ids += st[SB_U_OLD]; sig += 128 * st[SB_U_OLD]; n_sel = st[SB_U_NEW]; }   // captured chain: this batch's stored chunks
  const uint64_t sel = blockIdx.x;
  if (sel >= n_sel) return;
  const uint32_t t = threadIdx.x, lane = lane_id(), w = t >> 6;
  const uint64_t c = chunk_ids ? chunk_ids[sel] : sel;
  const uint64_t start = cuts[c];
  const uint64_t len = cuts[c + 1] - start;
  const uint64_t nsh = len >= 4 ? len - 3 : 0;
  const uint32_t sr0 = rotl32(seed_base + lane, 13), sr1 = rotl32(seed_base + lane + 64, 13);
  uint32_t m0 = 0xFFFFFFFFu, m1 = 0xFFFFFFFFu;
  constexpr int PART = MH_SLOTS / (NT / 64);  // each wavefront compacts and streams its own part of the table
  uint32_t* q = s_tab + w * PART;

  for (uint64_t sub0 = 0; sub0 < nsh; sub0 += MH_SUB) {
    const uint32_t cnt = (uint32_t)((nsh - sub0) < (uint64_t)MH_SUB ? (nsh - sub0) : (uint64_t)MH_SUB);
    // clear the set
    for (uint32_t i = t; i < MH_SLOTS / 4; i += NT) ((uint4*)s_tab)[i] = make_uint4(MH_EMPTY, MH_EMPTY, MH_EMPTY, MH_EMPTY);
    if (t == 0) s_flag = 0;
    __syncthreads();
    // insert R of every shingle of this pass
    const uint8_t* src = data + start + sub0;
    for (uint32_t p = t; p < (V == 3 ? 0u : cnt); p += NT) {
      const uint32_t R = murmur_R(load_u32_unaligned(src + p));
      if (R == MH_EMPTY) {
        s_flag = 1;
      } else {
        uint32_t slot = R >> (32 - MH_TBITS);
        for (;;) {
          const uint32_t old = atomicCAS(&s_tab[slot], MH_EMPTY, R);
          if (old == MH_EMPTY || old == R) break;
          slot = (slot + 1) & (MH_SLOTS - 1);
        }
      }
    }
    __syncthreads();
    // in-place compaction of this wave's part (write index never passes the read index)
    uint32_t wr = 0;
    for (uint32_t i = 0; i < (uint32_t)PART; i += 64) {
      const uint32_t v = q[i + lane];
      const bool valid = v != MH_EMPTY;
      const uint64_t m = __ballot(valid);
      if (valid) q[wr + (uint32_t)__builtin_popcountll(m & lanemask_lt())] = v;
      wr += (uint32_t)__builtin_popcountll(m);
    }
    // stream the distinct R values through the 2 seeds of this lane
    const uint32_t wr4 = V == 2 ? 0u : (wr & ~3u);
    if (V == 2) wr = 0;
    for (uint32_t i = 0; i < wr4; i += 4) {
      const uint4 kv = *(const uint4*)(q + i);
      const uint32_t a0 = murmur_tail<V>(kv.x, sr0), a1 = murmur_tail<V>(kv.x, sr1);
      const uint32_t b0 = murmur_tail<V>(kv.y, sr0), b1 = murmur_tail<V>(kv.y, sr1);
      const uint32_t c0 = murmur_tail<V>(kv.z, sr0), c1 = murmur_tail<V>(kv.z, sr1);
      const uint32_t d0 = murmur_tail<V>(kv.w, sr0), d1 = murmur_tail<V>(kv.w, sr1);
      m0 = min3u(min3u(m0, a0, b0), c0, d0);
      m1 = min3u(min3u(m1, a1, b1), c1, d1);
    }
    for (uint32_t i = wr4; i < wr; i++) {
      const uint32_t R = q[i];
      m0 = min(m0, murmur_tail<V>(R, sr0));
      m1 = min(m1, murmur_tail<V>(R, sr1));
    }
    if (w == 0 && s_flag) {  // the one R value that collides with the empty marker
      m0 = min(m0, murmur_tail<V>(MH_EMPTY, sr0));
      m1 = min(m1, murmur_tail<V>(MH_EMPTY, sr1));
    }
    __syncthreads();
  }
  s_sig[w][lane] = m0;
  s_sig[w][lane + 64] = m1;
  __syncthreads();
  if (t < 128) {
    uint32_t v = s_sig[0][t];
#pragma unroll
    for (int i = 1; i < NT / 64; i++) v = min(v, s_sig[i][t]);
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
  const dim3 grid((uint32_t)n_sel);
  PROF_BEGIN(HMSE_STAGE_L4_MINHASH, stream);
#ifdef HMSE_DIAG
  // diagnostic build only (libhmse_hip_diag.so, tools/minhash_variants.py): threads per workgroup x instruction variant;
  // variants 6 and 7 are timing probes that return WRONG signatures, which is why none of this is in the product library
  static const int variant = getenv("HMSE_MH_VARIANT") ? atoi(getenv("HMSE_MH_VARIANT")) : MH_DEFAULT_VARIANT;
  switch (variant) {
    case 0: l4_minhash_kernel<256, 0><<<grid, dim3(256), 0, stream>>>(data, n, cuts, chunk_ids, n_sel, cfg->seed_base, sig, nullptr); break;
    case 1: l4_minhash_kernel<256, 1><<<grid, dim3(256), 0, stream>>>(data, n, cuts, chunk_ids, n_sel, cfg->seed_base, sig, nullptr); break;
    case 2: l4_minhash_kernel<512, 0><<<grid, dim3(512), 0, stream>>>(data, n, cuts, chunk_ids, n_sel, cfg->seed_base, sig, nullptr); break;
    case 3: l4_minhash_kernel<512, 1><<<grid, dim3(512), 0, stream>>>(data, n, cuts, chunk_ids, n_sel, cfg->seed_base, sig, nullptr); break;
    case 4: l4_minhash_kernel<1024, 0><<<grid, dim3(1024), 0, stream>>>(data, n, cuts, chunk_ids, n_sel, cfg->seed_base, sig, nullptr); break;
    case 6: l4_minhash_kernel<1024, 2><<<grid, dim3(1024), 0, stream>>>(data, n, cuts, chunk_ids, n_sel, cfg->seed_base, sig, nullptr); break;  // timing probe: no tail
    case 7: l4_minhash_kernel<1024, 3><<<grid, dim3(1024), 0, stream>>>(data, n, cuts, chunk_ids, n_sel, cfg->seed_base, sig, nullptr); break;  // timing probe: no inserts
    default: l4_minhash_kernel<1024, 1><<<grid, dim3(1024), 0, stream>>>(data, n, cuts, chunk_ids, n_sel, cfg->seed_base, sig, nullptr); break;
  }
#else
  l4_minhash_kernel<1024, 1><<<grid, dim3(1024), 0, stream>>>(data, n, cuts, chunk_ids, n_sel, cfg->seed_base, sig, nullptr);
#endif
  PROF_END(HMSE_STAGE_L4_MINHASH, stream);
  HMSE_LAUNCH_CHECK();
  return HMSE_OK;
}

// captured chain: one workgroup per POSSIBLE stored chunk of the batch; those beyond the device-side count leave at once
int hmse_l4_minhash_dyn(const uint8_t* data, uint64_t n_cap, const uint64_t* cuts_all, const uint64_t* uniq_all, uint32_t* sig_all,
                        const uint64_t* st, uint64_t cap_chunks, const hmse_cfg* cfg, hipStream_t stream) {
  if (!data || !cuts_all || !uniq_all || !sig_all || !st || cap_chunks == 0 || cap_chunks > 0x7FFFFFFFull) return HMSE_EINVAL;
  l4_minhash_kernel<1024, 1><<<dim3((uint32_t)cap_chunks), dim3(1024), 0, stream>>>(data, n_cap, cuts_all, uniq_all, 0, cfg->seed_base, sig_all, st);
  HMSE_LAUNCH_CHECK();
  return HMSE_OK;
}
