// l3_sha256.hip — L3 per-chunk SHA-256 (FIPS 180-4) for gfx950.
//
// Replaces mbedtls_sha256(data, len, hash, 0) (README.md:2543) over every chunk.
// The compression function is a 64-step serial chain and Merkle-Damgard chains the blocks of a
// chunk, so the only parallel axis is ACROSS chunks: one chunk per LANE (64 chunks per wavefront,
// "one chunk per wavefront" would idle >= 48 lanes — SURVEY.md §7).  Lanes are persistent: a lane
// that finishes its chunk pulls the next chunk index from a device counter (wave-aggregated
// atomic), so the 2..32 KiB length skew of FastCDC chunks does not idle a wavefront's lanes.
// Blocks are fetched with four unaligned 16-B loads per lane (the HSA ABI enables unaligned
// global access); byte order is fixed with v_perm (bswap).
#include "common.h"

__constant__ uint32_t kK256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

// 3-input xor in one VALU op (v_bitop3_b32, truth table 0x96)
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return (uint32_t)__builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }

__device__ __forceinline__ void sha256_compress(uint32_t st[8], uint32_t w[16]) {
  uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
  for (int i = 0; i < 64; i++) {
    uint32_t wi;
    if (i < 16) {
      wi = w[i];
    } else {
      const uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
      const uint32_t s0 = xor3(rotr32(w15, 7), rotr32(w15, 18), w15 >> 3);
      const uint32_t s1 = xor3(rotr32(w2, 17), rotr32(w2, 19), w2 >> 10);
      wi = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
      w[i & 15] = wi;
    }
    const uint32_t S1 = xor3(rotr32(e, 6), rotr32(e, 11), rotr32(e, 25));
    const uint32_t ch = (uint32_t)__builtin_amdgcn_bitop3_b32(e, f, g, 0xCA);  // e ? f : g
    const uint32_t t1 = h + S1 + ch + kK256[i] + wi;
    const uint32_t S0 = xor3(rotr32(a, 2), rotr32(a, 13), rotr32(a, 22));
    const uint32_t mj = (uint32_t)__builtin_amdgcn_bitop3_b32(a, b, c, 0xE8);  // majority
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + S0 + mj;
  }
  st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

__global__ __launch_bounds__(256) void l3_sha256_kernel(const uint8_t* __restrict__ data, uint64_t n,
                                                         const uint64_t* __restrict__ cuts, uint64_t n_chunks,
                                                         uint8_t* __restrict__ digests, unsigned long long* counter,
                                                         const uint32_t* __restrict__ order, const uint64_t* __restrict__ sbst,
                                                         uint32_t batch_relative) {
  // captured chain: this batch's chunks; their digests go behind the earlier batches', or (batch_relative) into a per-batch buffer
  if (sbst) { cuts += sbst[SB_N_OLD]; if (!batch_relative) digests += 32 * sbst[SB_N_OLD]; n_chunks = sbst[SB_N_NEW]; }
  const uint32_t lane = lane_id();
  // per-lane chunk state
  uint64_t idx = 0, cur = 0, len = 0;
  uint32_t rem = 0;       // message bytes not yet consumed
  uint32_t phase = 3;     // 0 data blocks, 1 pad-only block pending, 3 idle (needs a chunk), 4 retired
  uint32_t st[8];
  uint64_t nx_addr = ~0ull;  // address of the prefetched block (none)
  uint4 nx0 = make_uint4(0, 0, 0, 0), nx1 = nx0, nx2 = nx0, nx3 = nx0;
  for (;;) {
    // ---- refill idle lanes (wave-aggregated fetch-add) ----
    const uint64_t need = __ballot(phase == 3);
    if (need) {
      const uint32_t leader = (uint32_t)__builtin_ctzll(need);
      unsigned long long base = 0;
      if (lane == leader) base = atomicAdd(counter, (unsigned long long)__builtin_popcountll(need));
      base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(base >> 32), (int)leader) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)base, (int)leader);
      if (phase == 3) {
        const uint64_t slot = base + (uint64_t)__builtin_popcountll(need & lanemask_lt());
        if (slot < n_chunks) {
          idx = order ? (uint64_t)order[slot] : slot;  // longest chunks are handed out first
          const uint64_t c0 = cuts[idx], c1 = cuts[idx + 1];
          cur = c0; len = c1 - c0; rem = (uint32_t)len; phase = 0;
          st[0] = 0x6a09e667; st[1] = 0xbb67ae85; st[2] = 0x3c6ef372; st[3] = 0xa54ff53a;
          st[4] = 0x510e527f; st[5] = 0x9b05688c; st[6] = 0x1f83d9ab; st[7] = 0x5be0cd19;
        } else {
          phase = 4;
        }
      }
    }
    if (__ballot(phase != 4) == 0) break;
    if (phase == 4) continue;  // retired lanes idle until the wave drains (all lanes re-converge at the ballots)

    // ---- build one 64-byte block ----
    uint32_t w[16];
    bool last;
    if (phase == 0) {
      if (nx_addr == cur) {  // block prefetched during the previous compression
        w[0] = nx0.x; w[1] = nx0.y; w[2] = nx0.z; w[3] = nx0.w; w[4] = nx1.x; w[5] = nx1.y; w[6] = nx1.z; w[7] = nx1.w;
        w[8] = nx2.x; w[9] = nx2.y; w[10] = nx2.z; w[11] = nx2.w; w[12] = nx3.x; w[13] = nx3.y; w[14] = nx3.z; w[15] = nx3.w;
      } else if (cur + 64 <= n) {
        const uint4 v0 = load_u4_unaligned(data + cur), v1 = load_u4_unaligned(data + cur + 16);
        const uint4 v2 = load_u4_unaligned(data + cur + 32), v3 = load_u4_unaligned(data + cur + 48);
        w[0] = v0.x; w[1] = v0.y; w[2] = v0.z; w[3] = v0.w; w[4] = v1.x; w[5] = v1.y; w[6] = v1.z; w[7] = v1.w;
        w[8] = v2.x; w[9] = v2.y; w[10] = v2.z; w[11] = v2.w; w[12] = v3.x; w[13] = v3.y; w[14] = v3.z; w[15] = v3.w;
      } else {  // within 64 bytes of the end of the buffer: bounded byte loads
#pragma unroll
        for (int i = 0; i < 16; i++) {
          uint32_t x = 0;
#pragma unroll
          for (int b = 0; b < 4; b++) {
            const uint64_t p = cur + 4 * i + b;
            if (p < n) x |= (uint32_t)data[p] << (8 * b);
          }
          w[i] = x;
        }
      }
      // prefetch the following block of this chunk; its latency hides under the 64 rounds below
      if (rem > 64 && cur + 128 <= n) {
        nx_addr = cur + 64;
        nx0 = load_u4_unaligned(data + nx_addr); nx1 = load_u4_unaligned(data + nx_addr + 16);
        nx2 = load_u4_unaligned(data + nx_addr + 32); nx3 = load_u4_unaligned(data + nx_addr + 48);
      }
#pragma unroll
      for (int i = 0; i < 16; i++) w[i] = __builtin_bswap32(w[i]);
      if (rem >= 64) {
        rem -= 64; cur += 64; last = false;
      } else {
        // final data block: keep `rem` message bytes, append 0x80, zero the rest
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const uint32_t lo = 4 * i;
          if (rem <= lo) w[i] = (rem == lo) ? 0x80000000u : 0u;
          else if (rem < lo + 4) {
            const uint32_t k = rem - lo;  // 1..3 valid (high-order) bytes
            w[i] = (w[i] & ~(0xFFFFFFFFu >> (8 * k))) | (0x80u << (24 - 8 * k));
          }
        }
        if (rem < 56) {
          const uint64_t bits = len * 8;
          w[14] = (uint32_t)(bits >> 32); w[15] = (uint32_t)bits; last = true;
        } else {
          phase = 1; last = false;
        }
        rem = 0;
      }
    } else {  // phase 1: pad-only block carrying the length
#pragma unroll
      for (int i = 0; i < 14; i++) w[i] = 0;
      const uint64_t bits = len * 8;
      w[14] = (uint32_t)(bits >> 32); w[15] = (uint32_t)bits; last = true;
    }
    sha256_compress(st, w);
    if (last) {
      uint4 o0, o1;
      o0.x = __builtin_bswap32(st[0]); o0.y = __builtin_bswap32(st[1]); o0.z = __builtin_bswap32(st[2]); o0.w = __builtin_bswap32(st[3]);
      o1.x = __builtin_bswap32(st[4]); o1.y = __builtin_bswap32(st[5]); o1.z = __builtin_bswap32(st[6]); o1.w = __builtin_bswap32(st[7]);
      uint4* dst = (uint4*)(digests + 32 * idx);
      __builtin_memcpy(dst, &o0, 16);
      __builtin_memcpy(dst + 1, &o1, 16);
      phase = 3;
    }
  }
}

// ---- hand-out order: chunks sorted by descending block count (counting sort, 1024 bins) -----------------
// Lanes hash one chunk each, so a wavefront runs as long as its longest chunk.  Handing chunks out longest
// first gives every wavefront 64 chunks of (nearly) equal length and leaves the short ones for the end,
// where they even out the finish times (longest-processing-time-first scheduling).
constexpr uint32_t ORD_BINS = 1024;
__device__ __forceinline__ uint32_t ord_key(uint64_t len) {
  const uint64_t nb = (len + 9 + 63) / 64;
  return ORD_BINS - 1 - (uint32_t)(nb < ORD_BINS - 1 ? nb : ORD_BINS - 1);  // descending length
}
__global__ __launch_bounds__(256) void ord_hist_kernel(const uint64_t* __restrict__ cuts, uint64_t n, uint32_t* __restrict__ bins,
                                                        const uint64_t* __restrict__ st) {
  if (st) { cuts += st[SB_N_OLD]; n = st[SB_N_NEW]; }
  __shared__ uint32_t s_bins[ORD_BINS];
  for (uint32_t i = threadIdx.x; i < ORD_BINS; i += 256) s_bins[i] = 0;
  __syncthreads();
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
    atomicAdd(&s_bins[ord_key(cuts[i + 1] - cuts[i])], 1u);
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < ORD_BINS; i += 256) if (s_bins[i]) atomicAdd(&bins[i], s_bins[i]);
}
__global__ __launch_bounds__(ORD_BINS) void ord_scan_kernel(uint32_t* __restrict__ bins) {
  __shared__ uint32_t s_red[ORD_BINS / 64 + 1];
  uint32_t total;
  const uint32_t v = bins[threadIdx.x];
  bins[threadIdx.x] = block_exclusive_scan<ORD_BINS>(v, s_red, &total);
}
// (block-aggregated: the chunk sizes of a corpus cluster in a few bins, one global atomic per chunk on those bins serialised the
// kernel — 134 us for the 107 k chunks of a 1 GB shard; now one global atomic per bin and workgroup)
__global__ __launch_bounds__(256) void ord_scatter_kernel(const uint64_t* __restrict__ cuts, uint64_t n, uint32_t* __restrict__ bins,
                                                           uint32_t* __restrict__ order, const uint64_t* __restrict__ st) {
  if (st) { cuts += st[SB_N_OLD]; n = st[SB_N_NEW]; }
  __shared__ uint32_t s_cnt[ORD_BINS];
  for (uint32_t b = threadIdx.x; b < ORD_BINS; b += 256) s_cnt[b] = 0;
  __syncthreads();
  constexpr uint32_t PER = 8;   // chunks per thread
  const uint64_t i0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * PER;
  uint32_t key[PER], rank[PER];
#pragma unroll
  for (uint32_t u = 0; u < PER; u++) {
    const uint64_t i = i0 + u;
    key[u] = i < n ? ord_key(cuts[i + 1] - cuts[i]) : 0xFFFFFFFFu;
    rank[u] = i < n ? atomicAdd(&s_cnt[key[u]], 1u) : 0u;
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < ORD_BINS; b += 256) { const uint32_t c = s_cnt[b]; if (c) s_cnt[b] = atomicAdd(&bins[b], c); }   // count -> base of this block's run
  __syncthreads();
#pragma unroll
  for (uint32_t u = 0; u < PER; u++) if (key[u] != 0xFFFFFFFFu) order[s_cnt[key[u]] + rank[u]] = (uint32_t)(i0 + u);
}

size_t hmse_l3_sha256_workspace_bytes_impl(uint64_t n_chunks) { return 256 + ORD_BINS * 4 + hmse_align_up((size_t)n_chunks * 4, 256); }

extern "C" int hmse_l3_sha256(const uint8_t* data, uint64_t n, const uint64_t* cuts, uint64_t n_chunks, uint8_t* digests,
                              void* ws, size_t ws_bytes, void* stream_) {
  if (n_chunks == 0) return HMSE_OK;
  if (!data || !cuts || !digests) return HMSE_EINVAL;
  if (!ws || ws_bytes < 8) return HMSE_ENOSPC;
  if (n_chunks >= 0xFFFFFFFFull) return HMSE_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  (void)hipGetLastError();  // drop stale errors of earlier runtime calls made by the host process
  // with enough workspace (hmse_workspace_bytes) chunks are handed out longest first; otherwise in index order
  uint32_t* bins = (uint32_t*)((uint8_t*)ws + 256);
  uint32_t* order = nullptr;
  if (ws_bytes >= hmse_l3_sha256_workspace_bytes_impl(n_chunks) && n_chunks > 4096) {
    order = bins + ORD_BINS;
    HMSE_FILL(ws, 0, 256 + ORD_BINS * 4, stream);
    uint64_t hb = (n_chunks + 255) / 256;
    if (hb > 1024) hb = 1024;
    ord_hist_kernel<<<dim3((uint32_t)hb), dim3(256), 0, stream>>>(cuts, n_chunks, bins, nullptr);
    ord_scan_kernel<<<dim3(1), dim3(ORD_BINS), 0, stream>>>(bins);
    ord_scatter_kernel<<<dim3((uint32_t)((n_chunks + 2047) / 2048)), dim3(256), 0, stream>>>(cuts, n_chunks, bins, order, nullptr);
    HMSE_LAUNCH_CHECK();
  } else {
    HMSE_FILL(ws, 0, 8, stream);
  }
  // persistent grid: one chunk per lane while chunks are scarce (the longest chunk bounds the run time
  // anyway), capped at 4 workgroups of 256 per CU (4 waves per SIMD saturate the VALU) so that on large
  // inputs every lane hashes several chunks and the dynamic hand-out evens out the 2..32 KiB length skew
  uint64_t blocks = (n_chunks + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  PROF_BEGIN(HMSE_STAGE_L3_SHA256, stream);
  l3_sha256_kernel<<<dim3((uint32_t)blocks), dim3(256), 0, stream>>>(data, n, cuts, n_chunks, digests, (unsigned long long*)ws, order, nullptr, 0u);
  PROF_END(HMSE_STAGE_L3_SHA256, stream);
  HMSE_LAUNCH_CHECK();
  return HMSE_OK;
}

// Captured chain: the chunks [st[SB_N_OLD], + st[SB_N_NEW]) of cuts_all, counts read on the device, grids sized for cap_chunks.
// Always hands chunks out longest first (same digests either way: the order only schedules).
int hmse_l3_sha256_dyn(const uint8_t* data, uint64_t n_cap, const uint64_t* cuts_all, uint8_t* digests_all, uint8_t* digests_batch,
                       const uint64_t* st, uint64_t cap_chunks, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (!data || !cuts_all || (!digests_all && !digests_batch) || !st || cap_chunks == 0 || cap_chunks >= 0xFFFFFFFFull) return HMSE_EINVAL;
  if (!ws || ws_bytes < hmse_l3_sha256_workspace_bytes_impl(cap_chunks)) return HMSE_ENOSPC;
  uint32_t* bins = (uint32_t*)((uint8_t*)ws + 256);
  uint32_t* order = bins + ORD_BINS;
  HMSE_FILL(ws, 0, 256 + ORD_BINS * 4, stream);
  uint64_t hb = (cap_chunks + 255) / 256;
  if (hb > 1024) hb = 1024;
  ord_hist_kernel<<<dim3((uint32_t)hb), dim3(256), 0, stream>>>(cuts_all, 0, bins, st);
  ord_scan_kernel<<<dim3(1), dim3(ORD_BINS), 0, stream>>>(bins);
  ord_scatter_kernel<<<dim3((uint32_t)((cap_chunks + 2047) / 2048)), dim3(256), 0, stream>>>(cuts_all, 0, bins, order, st);
  uint64_t blocks = (cap_chunks + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  l3_sha256_kernel<<<dim3((uint32_t)blocks), dim3(256), 0, stream>>>(data, n_cap, cuts_all, 0, digests_batch ? digests_batch : digests_all, (unsigned long long*)ws, order, st,
                                                                      digests_batch ? 1u : 0u);
  HMSE_LAUNCH_CHECK();
  return HMSE_OK;
}
