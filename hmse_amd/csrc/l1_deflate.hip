// l1_deflate.hip — L1 per-chunk DEFLATE (RFC 1951, raw) with a preset dictionary, for gfx950.
//
// Replaces mz_deflateInit2(&s, 9, MZ_DEFLATED, 15, 9, MZ_DEFAULT_STRATEGY) + mz_deflate(MZ_FINISH)
// (README.md:2374, 2378) applied per chunk, with the LSH base chunk as dictionary for the delta
// step (README.md:2160-2216 as resolved by SURVEY.md D6/D7).  A position-serial lazy matcher like
// zlib's cannot run wide, so the encoder is defined for parallel hardware (same definition as the
// CPU oracle, which it must match byte for byte):
//   1. 4-byte hash (12 bits) of every window position; positions bucket-sorted (counting sort by
//      LDS atomics + in-bucket rank) so that a position's candidates are its D predecessors in
//      its bucket: no pointer chasing, contiguous reads;
//   2. every chunk position finds its longest match in parallel (one lane per position);
//   3. the one-step-lazy parse p -> next(p) is a functional graph: reachable positions are marked
//      by pointer doubling (log rounds) instead of a serial walk;
//   4. histograms by LDS atomics, length-limited Huffman (sorted two-queue merge + Kraft repair),
//      closed-form code-length RLE, smallest of stored/fixed/dynamic;
//   5. token bit offsets by a workgroup prefix scan, bits OR-ed into an LDS image, coalesced copy-out.
// One workgroup per job (chunk, or chunk+dictionary); persistent workgroups pull jobs from a counter.
#include "common.h"

namespace dfl {

constexpr int HB = 12;
constexpr int NBK = 1 << HB;
constexpr uint32_t MINM = 4, MAXM = 258, WMAX = 32768;
constexpr int NT = 512;             // threads per workgroup
constexpr int WCAP = 65536;         // max dictionary + chunk
constexpr int LCAP = 32768;         // max chunk

__device__ __forceinline__ uint32_t hash4(uint32_t x) { return (x * 0x9E3779B1u) >> (32 - HB); }
__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

// RFC 1951 §3.2.5 closed forms
__device__ __forceinline__ void len_sym(uint32_t len, uint32_t& code, uint32_t& eb, uint32_t& ev) {
  if (len == 258) { code = 285; eb = 0; ev = 0; return; }
  const uint32_t l = len - 3;
  if (l < 8) { code = 257 + l; eb = 0; ev = 0; return; }
  const uint32_t e = (31 - __builtin_clz(l)) - 2;
  code = 257 + 4 * e + 4 + ((l >> e) & 3); eb = e; ev = l & ((1u << e) - 1);
}
__device__ __forceinline__ void dist_sym(uint32_t dist, uint32_t& code, uint32_t& eb, uint32_t& ev) {
  const uint32_t d = dist - 1;
  if (d < 4) { code = d; eb = 0; ev = 0; return; }
  const uint32_t hb = 31 - __builtin_clz(d), e = hb - 1;
  code = 2 * hb + ((d >> e) & 1); eb = e; ev = d & ((1u << e) - 1);
}
__device__ __forceinline__ uint32_t len_extra_bits(uint32_t code) {  // code 257..285
  if (code < 265 || code == 285) return 0;
  return (code - 261) >> 2;
}
__device__ __forceinline__ uint32_t dist_extra_bits(uint32_t code) { return code < 4 ? 0 : (code >> 1) - 1; }
__device__ __forceinline__ uint32_t fixed_len(uint32_t s) { return s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8; }

// order LDS traffic between the lanes of one wavefront (no instruction is emitted for the barrier itself)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

struct HuffScratch {
  uint32_t key[288];     // compacted (freq << 9 | sym)
  uint32_t sorted[288];  // ascending
  uint32_t w[576];       // node weights: leaves then internals
  uint16_t parent[576];
  uint32_t cnt[32];      // codes per length
  uint32_t m;            // used symbols
};

struct Shared {
  __attribute__((aligned(16))) uint8_t W[WCAP + 32];
  __attribute__((aligned(16))) uint32_t out[(LCAP + 64) / 4];
  uint32_t cur[NBK];
  uint32_t mark[LCAP / 32 + 2];
  uint32_t lf[288], df[32], cf[20];
  uint8_t ll[288], dl[32], cl[20];
  uint16_t lc[288], dc[32], cc[20];
  HuffScratch hs[2];
  uint8_t rle_sym[352], rle_eb[352], rle_ev[352];
  uint32_t red[NT / 64 + 1];
  uint32_t nr, nlit, ndist, ncl, mode, hdr_bits, total_bits;
  unsigned long long job;
};

// ---- wave-level Huffman length construction (one wavefront, mirrors oracle huff_lengths) --------
__device__ void huff_lengths_wave(uint32_t* freq, uint32_t n, uint32_t limit, uint8_t* lens, HuffScratch* hs) {
  const uint32_t lane = lane_id();
  // (1) at least two used symbols: the lowest unused indices get frequency 1
  uint32_t used = 0;
  for (uint32_t b = 0; b < n; b += 64) {
    const uint32_t s = b + lane;
    used += (uint32_t)__builtin_popcountll(__ballot(s < n && freq[s] != 0));
  }
  // (the histogram itself is left untouched: dummies only exist inside the tree construction)
  uint32_t dummy0 = 0xFFFFFFFFu, dummy1 = 0xFFFFFFFFu;
  if (used == 0) { dummy0 = 0; dummy1 = 1; }
  else if (used == 1) dummy0 = freq[0] != 0 ? 1u : 0u;
  // (2) compact used symbols and clear lens
  uint32_t m = 0;
  for (uint32_t b = 0; b < n; b += 64) {
    const uint32_t s = b + lane;
    uint32_t f = s < n ? freq[s] : 0u;
    if (s == dummy0 || s == dummy1) f = 1;
    if (s < n) lens[s] = 0;
    const bool u = f != 0;
    const uint64_t mask = __ballot(u);
    if (u) hs->key[m + (uint32_t)__builtin_popcountll(mask & lanemask_lt())] = (f << 9) | s;
    m += (uint32_t)__builtin_popcountll(mask);
  }
  wave_sync();
  // (3) rank sort ascending by (freq, sym)
  for (uint32_t j = lane; j < m; j += 64) {
    const uint32_t kj = hs->key[j];
    uint32_t r = 0;
    for (uint32_t i = 0; i < m; i++) r += hs->key[i] < kj;
    hs->sorted[r] = kj;
    hs->w[r] = kj >> 9;
  }
  if (lane < 32) hs->cnt[lane] = 0;
  wave_sync();
  // (4) two-queue merge (leaf wins ties), serial on lane 0
  if (lane == 0) {
    uint32_t li = 0, ii = m, ni = m;
    for (uint32_t step = 0; step + 1 < m; step++) {
      uint32_t pick0, pick1;
      {
        bool tl = (li < m && ii < ni) ? (hs->w[li] <= hs->w[ii]) : (li < m);
        pick0 = tl ? li++ : ii++;
      }
      {
        bool tl = (li < m && ii < ni) ? (hs->w[li] <= hs->w[ii]) : (li < m);
        pick1 = tl ? li++ : ii++;
      }
      hs->w[ni] = hs->w[pick0] + hs->w[pick1];
      hs->parent[pick0] = (uint16_t)ni; hs->parent[pick1] = (uint16_t)ni;
      ni++;
    }
    hs->m = m;
  }
  wave_sync();
  // (5) leaf depths by walking to the root; histogram of depths clamped to the limit
  const uint32_t root = 2 * m - 2;
  for (uint32_t j = lane; j < m; j += 64) {
    uint32_t d = 0, v = j;
    while (v != root) { v = hs->parent[v]; d++; }
    if (d > limit) d = limit;
    atomicAdd(&hs->cnt[d], 1u);
  }
  wave_sync();
  // (6) Kraft repair (rare) — serial
  if (lane == 0) {
    uint32_t total = 0;
    for (uint32_t i = limit; i > 0; i--) total += hs->cnt[i] << (limit - i);
    while (total > (1u << limit)) {
      hs->cnt[limit]--;
      for (uint32_t i = limit - 1; i > 0; i--) if (hs->cnt[i]) { hs->cnt[i]--; hs->cnt[i + 1] += 2; break; }
      total--;
    }
  }
  wave_sync();
  // (7) hand lengths out in sorted order: the most frequent symbol gets the shortest length
  for (uint32_t j = lane; j < m; j += 64) {
    const uint32_t jt = m - 1 - j;  // rank from the most frequent
    uint32_t cum = 0, l = 0;
    for (uint32_t i = 1; i <= limit; i++) { cum += hs->cnt[i]; if (l == 0 && cum > jt) l = i; }
    lens[hs->sorted[j] & 511u] = (uint8_t)l;
  }
  wave_sync();
}

// canonical codes (bit-reversed for LSB-first packing); one wavefront
__device__ void huff_codes_wave(const uint8_t* lens, uint32_t n, uint16_t* codes, uint32_t* cnt /*>=32 u32 scratch*/) {
  const uint32_t lane = lane_id();
  if (lane < 32) cnt[lane] = 0;
  wave_sync();
  for (uint32_t s = lane; s < n; s += 64) if (lens[s]) atomicAdd(&cnt[lens[s]], 1u);
  wave_sync();
  uint32_t next[16];
  {
    uint32_t code = 0, prev = 0;
    next[0] = 0;
    for (uint32_t b = 1; b <= 15; b++) { code = (code + prev) << 1; next[b] = code; prev = cnt[b]; }
  }
  // a symbol's code = next[len] + (#earlier symbols with the same length)
  uint32_t run[16];
#pragma unroll
  for (int b = 0; b < 16; b++) run[b] = 0;
  for (uint32_t b0 = 0; b0 < n; b0 += 64) {
    const uint32_t s = b0 + lane;
    const uint32_t l = s < n ? lens[s] : 0u;
    uint32_t mycode = 0;
#pragma unroll
    for (uint32_t b = 1; b <= 15; b++) {
      const uint64_t mask = __ballot(l == b);
      if (l == b) mycode = next[b] + run[b] + (uint32_t)__builtin_popcountll(mask & lanemask_lt());
      run[b] += (uint32_t)__builtin_popcountll(mask);
    }
    if (s < n) codes[s] = l ? (uint16_t)(__builtin_bitreverse32(mycode) >> (32 - l)) : (uint16_t)0;
  }
  wave_sync();
}

__device__ __forceinline__ void put_bits(uint32_t* out, uint32_t off, uint32_t v, uint32_t nb) {
  if (nb == 0) return;
  const uint32_t w = off >> 5, s = off & 31;
  atomicOr(&out[w], v << s);
  if (s + nb > 32) atomicOr(&out[w + 1], v >> (32 - s));
}

struct Args {
  const uint8_t* data; uint64_t n;
  const uint64_t* cuts; const uint64_t* chunk_ids; const int64_t* base; uint64_t n_sel;
  uint32_t depth;
  const uint64_t* slot_off;  // [n_sel] byte offset of chunk k's slot pair
  uint8_t* slots;            // FULL stream at slot_off[k], DELTA stream at slot_off[k] + slot_stride(len)
  uint64_t slot_cap; uint32_t* status;
  uint32_t* len_full; uint32_t* len_delta;
  uint8_t* scratch; size_t scratch_stride;
  unsigned long long* counter;
};

__device__ __forceinline__ uint32_t slot_stride(uint32_t len) { return (len + 5 + 15) & ~15u; }

// per-workgroup global scratch layout
struct Scratch {
  uint16_t S1[WCAP]; uint16_t S[WCAP];
  uint16_t jumpA[LCAP + 8]; uint16_t jumpB[LCAP + 8];
  uint16_t mdist[LCAP]; uint8_t mlen[LCAP];
};

__global__ __launch_bounds__(NT) void l1_deflate_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
  Shared& sh = *reinterpret_cast<Shared*>(smem_raw);
  Scratch& sc = *reinterpret_cast<Scratch*>(a.scratch + (size_t)blockIdx.x * a.scratch_stride);
  const uint32_t t = threadIdx.x, lane = lane_id(), wave = t >> 6;

  for (;;) {
    __syncthreads();
    if (t == 0) sh.job = atomicAdd(a.counter, 1ull);
    __syncthreads();
    const unsigned long long job = sh.job;
    if (job >= 2 * a.n_sel) break;
    const uint64_t k = job >> 1;
    const uint32_t variant = (uint32_t)(job & 1);
    const int64_t bsel = a.base ? a.base[k] : -1;
    if (variant == 1 && bsel < 0) continue;
    const uint64_t c = a.chunk_ids ? a.chunk_ids[k] : k;
    const uint64_t cstart = a.cuts[c];
    const uint32_t L = (uint32_t)(a.cuts[c + 1] - cstart);
    uint32_t Dl = 0; uint64_t dstart = 0;
    if (variant == 1) {
      const uint64_t bc = a.chunk_ids ? a.chunk_ids[bsel] : (uint64_t)bsel;
      dstart = a.cuts[bc];
      uint64_t dl64 = a.cuts[bc + 1] - dstart;
      if (dl64 > WMAX) { dstart += dl64 - WMAX; dl64 = WMAX; }
      Dl = (uint32_t)dl64;
    }
    uint8_t* slot = a.slots + a.slot_off[k] + (variant ? slot_stride(L) : 0u);
    uint32_t* len_out = variant ? a.len_delta : a.len_full;
    if (L > LCAP || a.slot_off[k] + (variant + 1ull) * slot_stride(L) > a.slot_cap) {  // not encodable / no room: flagged
      if (t == 0) { len_out[k] = 0xFFFFFFFFu; if (L <= LCAP) atomicOr(a.status, 2u); }
      continue;
    }
    const uint32_t T = Dl + L;
    const uint8_t* csrc = a.data + cstart;
    const uint8_t* dsrc = a.data + dstart;

    // ---- phase 0: window into LDS, clear tables -----------------------------------------------
    for (uint32_t i = t * 16; i < Dl; i += NT * 16) {
      if (i + 16 <= Dl) { const uint4 v = load_u4_unaligned(dsrc + i); __builtin_memcpy(sh.W + i, &v, 16); }
      else for (uint32_t b = i; b < Dl; b++) sh.W[b] = dsrc[b];
    }
    for (uint32_t i = t * 16; i < L; i += NT * 16) {
      if (i + 16 <= L) { const uint4 v = load_u4_unaligned(csrc + i); __builtin_memcpy(sh.W + Dl + i, &v, 16); }
      else for (uint32_t b = i; b < L; b++) sh.W[Dl + b] = csrc[b];
    }
    if (t < 32) sh.W[T + t] = 0;
    for (uint32_t i = t; i < NBK; i += NT) sh.cur[i] = 0;
    for (uint32_t i = t; i < LCAP / 32 + 2; i += NT) sh.mark[i] = 0;
    for (uint32_t i = t; i < 288; i += NT) sh.lf[i] = 0;
    if (t < 32) sh.df[t] = 0;
    if (t < 20) sh.cf[t] = 0;
    for (uint32_t i = t; i < (L + 64) / 4; i += NT) sh.out[i] = 0;
    for (uint32_t i = t; i < L; i += NT) sc.mlen[i] = 0;
    __syncthreads();

    // ---- phase 1-2: bucket histogram + exclusive scan -> bucket starts --------------------------
    const uint32_t nh = T >= 4 ? T - 3 : 0;
    for (uint32_t q = t; q < nh; q += NT) atomicAdd(&sh.cur[hash4(ld32(sh.W + q))], 1u);
    __syncthreads();
    {
      constexpr int PER = NBK / NT;
      uint32_t loc[PER], sum = 0;
#pragma unroll
      for (int i = 0; i < PER; i++) { loc[i] = sh.cur[t * PER + i]; sum += loc[i]; }
      uint32_t total;
      uint32_t ex = block_exclusive_scan<NT>(sum, sh.red, &total);
#pragma unroll
      for (int i = 0; i < PER; i++) { sh.cur[t * PER + i] = ex; ex += loc[i]; }
    }
    __syncthreads();
    // ---- phase 3: scatter (order inside a bucket is arbitrary here) -----------------------------
    for (uint32_t q = t; q < nh; q += NT) {
      const uint32_t slot_i = atomicAdd(&sh.cur[hash4(ld32(sh.W + q))], 1u);
      sc.S1[slot_i] = (uint16_t)q;
    }
    __syncthreads();  // cur[h] now = end of bucket h
    // ---- phase 4: rank inside the bucket -> ascending positions ---------------------------------
    for (uint32_t i = t; i < nh; i += NT) {
      const uint32_t q = sc.S1[i];
      const uint32_t h = hash4(ld32(sh.W + q));
      const uint32_t lo = h ? sh.cur[h - 1] : 0u, hi = sh.cur[h];
      uint32_t r = 0;
      for (uint32_t j = lo; j < hi; j++) r += sc.S1[j] < q;
      sc.S[lo + r] = (uint16_t)q;
    }
    __syncthreads();
    // ---- phase 5: longest match for every chunk position ----------------------------------------
    for (uint32_t i = t; i < nh; i += NT) {
      const uint32_t p = sc.S[i];
      if (p < Dl) continue;
      const uint32_t h = hash4(ld32(sh.W + p));
      const uint32_t lo = h ? sh.cur[h - 1] : 0u;
      const uint32_t maxlen = (T - p) < MAXM ? (T - p) : MAXM;
      uint32_t best = MINM - 1, bd = 0;
      for (uint32_t kk = 1; kk <= a.depth && i >= lo + kk; kk++) {
        const uint32_t q = sc.S[i - kk];
        if (p - q > WMAX) break;
        if (sh.W[q + best] != sh.W[p + best]) continue;  // cannot beat the current best
        uint32_t ml = 0;
        while (ml < maxlen) {
          const uint32_t x = ld32(sh.W + q + ml) ^ ld32(sh.W + p + ml);
          if (x) { ml += (uint32_t)__builtin_ctz(x) >> 3; break; }
          ml += 4;
        }
        if (ml > maxlen) ml = maxlen;
        if (ml > best) { best = ml; bd = p - q; if (ml == maxlen) break; }
      }
      if (best >= MINM) { sc.mlen[p - Dl] = (uint8_t)(best - 3); sc.mdist[p - Dl] = (uint16_t)bd; }
    }
    __syncthreads();
    // ---- phase 6: parse by pointer doubling -------------------------------------------------------
    auto take = [&](uint32_t x) -> bool {
      const uint32_t ml = sc.mlen[x];
      return ml != 0 && !(x + 1 < L && sc.mlen[x + 1] > ml);
    };
    for (uint32_t x = t; x <= L; x += NT) {
      uint32_t nx = L;
      if (x < L) { nx = x + (take(x) ? (uint32_t)sc.mlen[x] + 3u : 1u); if (nx > L) nx = L; }
      sc.jumpA[x] = (uint16_t)nx;
    }
    if (t == 0 && L > 0) sh.mark[0] = 1u;
    __syncthreads();
    {
      uint16_t* ja = sc.jumpA; uint16_t* jb = sc.jumpB;
      for (int round = 0; round < 17; round++) {
        if (L == 0 || ja[0] == L) break;  // uniform: every chain position is already marked
        for (uint32_t x = t; x < L; x += NT) {
          if ((sh.mark[x >> 5] >> (x & 31)) & 1u) {
            const uint32_t j = ja[x];
            if (j < L) atomicOr(&sh.mark[j >> 5], 1u << (j & 31));
          }
          const uint32_t j1 = ja[x];
          jb[x] = j1 < L ? ja[j1] : (uint16_t)L;
        }
        if (t == 0) jb[L] = (uint16_t)L;
        __syncthreads();
        uint16_t* tmp = ja; ja = jb; jb = tmp;
      }
    }
    __syncthreads();
    // ---- phase 7: symbol histograms -----------------------------------------------------------------
    for (uint32_t x = t; x < L; x += NT) {
      if ((sh.mark[x >> 5] >> (x & 31)) & 1u) {
        if (take(x)) {
          uint32_t code, eb, ev;
          len_sym((uint32_t)sc.mlen[x] + 3u, code, eb, ev); atomicAdd(&sh.lf[code], 1u);
          dist_sym(sc.mdist[x], code, eb, ev); atomicAdd(&sh.df[code], 1u);
        } else atomicAdd(&sh.lf[sh.W[Dl + x]], 1u);
      }
    }
    if (t == 0) atomicAdd(&sh.lf[256], 1u);
    __syncthreads();
    // ---- phase 8: trees (wave 0: lit/len, wave 1: dist) ------------------------------------------------
    // fixed / stored costs first (histograms are modified by the >= 2 symbols rule)
    if (wave == 2) {
      uint32_t fb = 0, xb = 0;
      for (uint32_t s = lane; s < 286; s += 64) { fb += sh.lf[s] * fixed_len(s); if (s >= 257) xb += sh.lf[s] * len_extra_bits(s); }
      if (lane < 30) { fb += sh.df[lane] * 5u; xb += sh.df[lane] * dist_extra_bits(lane); }
      for (int d = 32; d > 0; d >>= 1) { fb += __shfl_down(fb, d, 64); xb += __shfl_down(xb, d, 64); }
      if (lane == 0) { sh.hdr_bits = fb + xb + 3; sh.total_bits = xb; }  // hdr_bits: fixed total, total_bits: extra bits (temporaries)
    }
    __syncthreads();
    if (wave == 0) huff_lengths_wave(sh.lf, 286, 15, sh.ll, &sh.hs[0]);
    if (wave == 1) huff_lengths_wave(sh.df, 30, 15, sh.dl, &sh.hs[1]);
    __syncthreads();
    // ---- phase 9: code-length RLE, CL tree, block type (wave 0) -------------------------------------------
    if (wave == 0) {
      if (lane == 0) {
        uint32_t nlit = 286; while (nlit > 257 && sh.ll[nlit - 1] == 0) nlit--;
        uint32_t ndist = 30; while (ndist > 1 && sh.dl[ndist - 1] == 0) ndist--;
        sh.nlit = nlit; sh.ndist = ndist;
        uint32_t kq = 0;
        for (int tree = 0; tree < 2; tree++) {
          const uint8_t* l = tree ? sh.dl : sh.ll;
          const uint32_t nn = tree ? ndist : nlit;
          uint32_t i = 0;
          while (i < nn) {
            uint32_t j = i + 1;
            while (j < nn && l[j] == l[i]) j++;
            uint32_t run = j - i; const uint32_t v = l[i];
            if (v == 0) {
              while (run >= 11) { const uint32_t cc = run > 138 ? 138 : run; sh.rle_sym[kq] = 18; sh.rle_eb[kq] = 7; sh.rle_ev[kq] = (uint8_t)(cc - 11); kq++; run -= cc; }
              if (run >= 3) { sh.rle_sym[kq] = 17; sh.rle_eb[kq] = 3; sh.rle_ev[kq] = (uint8_t)(run - 3); kq++; run = 0; }
              while (run) { sh.rle_sym[kq] = 0; sh.rle_eb[kq] = 0; sh.rle_ev[kq] = 0; kq++; run--; }
            } else {
              sh.rle_sym[kq] = (uint8_t)v; sh.rle_eb[kq] = 0; sh.rle_ev[kq] = 0; kq++; run--;
              while (run >= 3) { const uint32_t cc = run > 6 ? 6 : run; sh.rle_sym[kq] = 16; sh.rle_eb[kq] = 2; sh.rle_ev[kq] = (uint8_t)(cc - 3); kq++; run -= cc; }
              while (run) { sh.rle_sym[kq] = (uint8_t)v; sh.rle_eb[kq] = 0; sh.rle_ev[kq] = 0; kq++; run--; }
            }
            i = j;
          }
        }
        sh.nr = kq;
        for (uint32_t i = 0; i < kq; i++) sh.cf[sh.rle_sym[i]]++;
      }
      wave_sync();
      huff_lengths_wave(sh.cf, 19, 7, sh.cl, &sh.hs[0]);
      if (lane == 0) {
        const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint32_t ncl = 19; while (ncl > 4 && sh.cl[order[ncl - 1]] == 0) ncl--;
        sh.ncl = ncl;
        const uint32_t fixed_bits = sh.hdr_bits, extra = sh.total_bits;
        uint32_t dyn = 3 + 14 + 3 * ncl + extra;
        for (uint32_t i = 0; i < sh.nr; i++) dyn += sh.cl[sh.rle_sym[i]] + sh.rle_eb[i];
        uint32_t dhdr = dyn - extra;  // header bits so far (3 + 14 + 3*ncl + cl tokens)
        for (uint32_t s = 0; s < 286; s++) dyn += sh.lf[s] * sh.ll[s];
        for (uint32_t s = 0; s < 30; s++) dyn += sh.df[s] * sh.dl[s];
        const uint32_t stored = 8u * (5u + L);
        uint32_t mode;
        if (stored <= fixed_bits && stored <= dyn) mode = 0;
        else if (fixed_bits <= dyn) mode = 1;
        else mode = 2;
        sh.mode = mode;
        sh.hdr_bits = mode == 2 ? dhdr : 3u;
      }
    }
    __syncthreads();
    const uint32_t mode = sh.mode;
    if (mode == 0) {
      if (t == 0) {
        slot[0] = 1; slot[1] = (uint8_t)L; slot[2] = (uint8_t)(L >> 8); slot[3] = (uint8_t)~L; slot[4] = (uint8_t)(~L >> 8);
        len_out[k] = 5 + L;
      }
      for (uint32_t x = t; x < L; x += NT) slot[5 + x] = sh.W[Dl + x];
      continue;
    }
    // ---- phase 10: code tables ------------------------------------------------------------------------------
    if (mode == 1) {
      for (uint32_t s = t; s < 288; s += NT) sh.ll[s] = (uint8_t)fixed_len(s);
      if (t < 32) sh.dl[t] = 5;
      __syncthreads();
    }
    if (wave == 0) huff_codes_wave(sh.ll, mode == 1 ? 288 : 286, sh.lc, sh.hs[0].cnt);
    if (wave == 1) huff_codes_wave(sh.dl, mode == 1 ? 32 : 30, sh.dc, sh.hs[1].cnt);
    if (wave == 2 && mode == 2) huff_codes_wave(sh.cl, 19, sh.cc, sh.hs[1].key);
    __syncthreads();
    // ---- phase 11: emit ----------------------------------------------------------------------------------------
    if (t == 0) {
      uint32_t off = 0;
      put_bits(sh.out, off, 1, 1); off += 1;
      put_bits(sh.out, off, mode, 2); off += 2;
      if (mode == 2) {
        const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        put_bits(sh.out, off, sh.nlit - 257, 5); off += 5;
        put_bits(sh.out, off, sh.ndist - 1, 5); off += 5;
        put_bits(sh.out, off, sh.ncl - 4, 4); off += 4;
        for (uint32_t i = 0; i < sh.ncl; i++) { put_bits(sh.out, off, sh.cl[order[i]], 3); off += 3; }
        for (uint32_t i = 0; i < sh.nr; i++) {
          const uint32_t s = sh.rle_sym[i];
          put_bits(sh.out, off, sh.cc[s], sh.cl[s]); off += sh.cl[s];
          if (sh.rle_eb[i]) { put_bits(sh.out, off, sh.rle_ev[i], sh.rle_eb[i]); off += sh.rle_eb[i]; }
        }
      }
    }
    // token bits: each thread owns a contiguous block of positions
    const uint32_t per = (L + NT - 1) / NT;
    const uint32_t x0 = t * per, x1 = (x0 + per) < L ? (x0 + per) : L;
    uint32_t mybits = 0;
    for (uint32_t x = x0; x < x1; x++) {
      if ((sh.mark[x >> 5] >> (x & 31)) & 1u) {
        if (take(x)) {
          uint32_t code, eb, ev, dcode, deb, dev;
          len_sym((uint32_t)sc.mlen[x] + 3u, code, eb, ev);
          dist_sym(sc.mdist[x], dcode, deb, dev);
          mybits += sh.ll[code] + eb + sh.dl[dcode] + deb;
        } else mybits += sh.ll[sh.W[Dl + x]];
      }
    }
    uint32_t total;
    uint32_t off = block_exclusive_scan<NT>(mybits, sh.red, &total) + sh.hdr_bits;
    for (uint32_t x = x0; x < x1; x++) {
      if ((sh.mark[x >> 5] >> (x & 31)) & 1u) {
        if (take(x)) {
          uint32_t code, eb, ev, dcode, deb, dev;
          len_sym((uint32_t)sc.mlen[x] + 3u, code, eb, ev);
          dist_sym(sc.mdist[x], dcode, deb, dev);
          const uint32_t l1 = sh.ll[code], l2 = sh.dl[dcode];
          put_bits(sh.out, off, (uint32_t)sh.lc[code] | (ev << l1), l1 + eb); off += l1 + eb;
          put_bits(sh.out, off, (uint32_t)sh.dc[dcode] | (dev << l2), l2 + deb); off += l2 + deb;
        } else {
          const uint32_t b = sh.W[Dl + x];
          put_bits(sh.out, off, sh.lc[b], sh.ll[b]); off += sh.ll[b];
        }
      }
    }
    const uint32_t end_bits = sh.hdr_bits + total;
    if (t == 0) put_bits(sh.out, end_bits, sh.lc[256], sh.ll[256]);
    __syncthreads();
    const uint32_t nbytes = (end_bits + sh.ll[256] + 7) >> 3;
    for (uint32_t i = t * 16; i < nbytes; i += NT * 16) {  // slot is 16-byte aligned and padded
      const uint4 v = *(const uint4*)((const uint8_t*)sh.out + i);
      *(uint4*)(slot + i) = v;
    }
    if (t == 0) len_out[k] = nbytes;
  }
}

// slot sizes: FULL (+ DELTA when a base exists), 16-byte aligned; also clears the lengths
__global__ __launch_bounds__(256) void slot_size_kernel(const uint64_t* __restrict__ cuts, const uint64_t* __restrict__ chunk_ids,
                                                         const int64_t* __restrict__ base, uint64_t n_sel,
                                                         uint64_t* __restrict__ sizes) {
  const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_sel) return;
  const uint64_t c = chunk_ids ? chunk_ids[k] : k;
  const uint32_t len = (uint32_t)(cuts[c + 1] - cuts[c]);
  const uint32_t s = slot_stride(len);
  sizes[k] = (base && base[k] >= 0) ? 2ull * s : (uint64_t)s;
}

// kind decision (README.md:1328, 2175 as resolved by SURVEY.md D7) and final lengths
__global__ __launch_bounds__(256) void decide_kernel(const uint64_t* __restrict__ cuts, const uint64_t* __restrict__ chunk_ids,
                                                      const int64_t* __restrict__ base, uint64_t n_sel,
                                                      const uint32_t* __restrict__ len_full, const uint32_t* __restrict__ len_delta,
                                                      uint32_t pct, uint64_t* __restrict__ final_len, uint8_t* __restrict__ kind,
                                                      uint32_t* status) {
  const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_sel) return;
  const uint32_t n1 = len_full[k];
  uint32_t n = n1; uint8_t kd = HMSE_KIND_FULL;
  if (n1 == 0xFFFFFFFFu) { atomicOr(status, 4u); n = 0; }
  else if (base && base[k] >= 0) {
    const uint32_t n2 = len_delta[k];
    const uint64_t c = chunk_ids ? chunk_ids[k] : k;
    const uint64_t len = cuts[c + 1] - cuts[c];
    bool ok = n2 != 0xFFFFFFFFu && (uint64_t)n2 + 8 < n1;
    if (pct && (uint64_t)n2 * 100 > (uint64_t)pct * len) ok = false;
    if (ok) { n = n2; kd = HMSE_KIND_DELTA; }
  }
  final_len[k] = n;
  if (kind) kind[k] = kd;
}

__global__ __launch_bounds__(256) void gather_kernel(const uint64_t* __restrict__ cuts, const uint64_t* __restrict__ chunk_ids,
                                                      uint64_t n_sel, const uint64_t* __restrict__ slot_off,
                                                      const uint8_t* __restrict__ slots, const uint8_t* __restrict__ kind,
                                                      const uint64_t* __restrict__ out_off, uint8_t* __restrict__ out,
                                                      uint64_t out_cap, uint32_t* status) {
  const uint64_t k = blockIdx.x;
  if (k >= n_sel) return;
  const uint64_t c = chunk_ids ? chunk_ids[k] : k;
  const uint32_t len = (uint32_t)(cuts[c + 1] - cuts[c]);
  const uint8_t* src = slots + slot_off[k] + ((kind && kind[k] == HMSE_KIND_DELTA) ? slot_stride(len) : 0u);
  const uint64_t o0 = out_off[k], o1 = out_off[k + 1];
  if (o1 > out_cap) { if (threadIdx.x == 0) atomicOr(status, 1u); return; }
  const uint32_t nb = (uint32_t)(o1 - o0);
  uint8_t* dst = out + o0;
  for (uint32_t i = threadIdx.x * 16; i < nb; i += blockDim.x * 16) {
    if (i + 16 <= nb) { const uint4 v = *(const uint4*)(src + i); __builtin_memcpy(dst + i, &v, 16); }
    else for (uint32_t b = i; b < nb; b++) dst[b] = src[b];
  }
}

// ---- u64 exclusive scan (three small kernels) ---------------------------------------------------------------
constexpr int SC_NT = 1024;
__global__ __launch_bounds__(SC_NT) void scan_reduce_kernel(const uint64_t* __restrict__ in, uint64_t n, uint64_t* __restrict__ bsum) {
  __shared__ unsigned long long s[SC_NT / 64];
  const uint64_t i = (uint64_t)blockIdx.x * SC_NT + threadIdx.x;
  unsigned long long v = i < n ? in[i] : 0ull;
  for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
  if (lane_id() == 0) s[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) { unsigned long long tt = 0; for (int w = 0; w < SC_NT / 64; w++) tt += s[w]; bsum[blockIdx.x] = tt; }
}
__global__ __launch_bounds__(SC_NT) void scan_blocks_kernel(uint64_t* bsum, uint64_t nb, uint64_t* total_out) {
  __shared__ unsigned long long s[SC_NT / 64 + 1];
  __shared__ unsigned long long run;
  if (threadIdx.x == 0) run = 0;
  __syncthreads();
  for (uint64_t b0 = 0; b0 < nb; b0 += SC_NT) {
    const uint64_t i = b0 + threadIdx.x;
    const unsigned long long v = i < nb ? bsum[i] : 0ull;
    unsigned long long inc = v;
    for (int d = 1; d < 64; d <<= 1) { unsigned long long tt = __shfl_up(inc, d, 64); if (lane_id() >= (uint32_t)d) inc += tt; }
    if (lane_id() == 63) s[threadIdx.x >> 6] = inc;
    __syncthreads();
    unsigned long long wb = 0, tot = 0;
    for (int w = 0; w < SC_NT / 64; w++) { if ((uint32_t)w < (threadIdx.x >> 6)) wb += s[w]; tot += s[w]; }
    const unsigned long long r = run;
    if (i < nb) bsum[i] = r + wb + inc - v;
    __syncthreads();
    if (threadIdx.x == 0) run = r + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0 && total_out) *total_out = run;
}
__global__ __launch_bounds__(SC_NT) void scan_apply_kernel(const uint64_t* __restrict__ in, uint64_t n, const uint64_t* __restrict__ bsum,
                                                            uint64_t* __restrict__ out) {
  __shared__ unsigned long long s[SC_NT / 64];
  const uint64_t i = (uint64_t)blockIdx.x * SC_NT + threadIdx.x;
  const unsigned long long v = i < n ? in[i] : 0ull;
  unsigned long long inc = v;
  for (int d = 1; d < 64; d <<= 1) { unsigned long long tt = __shfl_up(inc, d, 64); if (lane_id() >= (uint32_t)d) inc += tt; }
  if (lane_id() == 63) s[threadIdx.x >> 6] = inc;
  __syncthreads();
  unsigned long long wb = 0;
  for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) wb += s[w];
  if (i < n) out[i] = bsum[blockIdx.x] + wb + inc - v;
}

// out[0..n) = exclusive scan of in; *total (device) = sum. in and out may alias.
static int exclusive_scan_u64(const uint64_t* in, uint64_t n, uint64_t* out, uint64_t* bsum, uint64_t* total, hipStream_t stream) {
  const uint64_t nb = (n + SC_NT - 1) / SC_NT;
  scan_reduce_kernel<<<dim3((uint32_t)nb), dim3(SC_NT), 0, stream>>>(in, n, bsum);
  scan_blocks_kernel<<<dim3(1), dim3(SC_NT), 0, stream>>>(bsum, nb, total);
  scan_apply_kernel<<<dim3((uint32_t)nb), dim3(SC_NT), 0, stream>>>(in, n, bsum, out);
  return hipGetLastError() == hipSuccess ? HMSE_OK : HMSE_EHIP;
}

constexpr int N_WG = 256;  // persistent workgroups (one per CU: the LDS image is > 80 KiB)

struct Ws {
  unsigned long long* counter; uint64_t* slot_off; uint64_t* final_len; uint64_t* bsum; uint64_t* slot_total;
  uint32_t* len_full; uint32_t* len_delta; uint8_t* scratch; uint8_t* slots; size_t fixed_bytes;
};
static Ws carve(void* ws, uint64_t n_sel) {
  WsCarver w(ws, ~(size_t)0);
  Ws r;
  r.counter = w.take<unsigned long long>(1);
  r.slot_total = w.take<uint64_t>(1);
  r.slot_off = w.take<uint64_t>(n_sel + 1);
  r.final_len = w.take<uint64_t>(n_sel + 1);
  r.bsum = w.take<uint64_t>((n_sel + SC_NT - 1) / SC_NT + 1);
  r.len_full = w.take<uint32_t>(n_sel);
  r.len_delta = w.take<uint32_t>(n_sel);
  r.scratch = w.take<uint8_t>((size_t)N_WG * hmse_align_up(sizeof(Scratch), 256));
  r.fixed_bytes = w.off;
  r.slots = r.scratch ? (uint8_t*)ws + w.off : nullptr;
  return r;
}

}  // namespace dfl

// fixed part only; the caller adds the slot area: sum over chunks of align16(len+5) * (1 + has_base)
size_t hmse_l1_deflate_workspace_bytes_impl(uint64_t n_sel, const hmse_cfg*) { return dfl::carve(nullptr, n_sel).fixed_bytes; }

extern "C" int hmse_l1_deflate(const uint8_t* data, uint64_t n, const uint64_t* cuts, const uint64_t* chunk_ids,
                               const int64_t* base, uint64_t n_sel, const hmse_cfg* cfg, uint8_t* out, uint64_t out_cap,
                               uint64_t* out_off, uint8_t* kind, uint32_t* status, void* ws, size_t ws_bytes, void* stream_) {
  using namespace dfl;
  if (hmse_cfg_validate_impl(cfg) != 0) return HMSE_EINVAL;
  if (!out_off || !status) return HMSE_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  (void)hipGetLastError();  // drop stale errors of earlier runtime calls made by the host process
  HMSE_HIP(hipMemsetAsync(status, 0, sizeof(uint32_t), stream));
  if (n_sel == 0) { HMSE_HIP(hipMemsetAsync(out_off, 0, sizeof(uint64_t), stream)); return HMSE_OK; }
  if (!data || !cuts || !out || !kind) return HMSE_EINVAL;
  if (n_sel > 0x3FFFFFFFull) return HMSE_EINVAL;
  Ws w = carve(ws, n_sel);
  if (!ws || ws_bytes < w.fixed_bytes) return HMSE_ENOSPC;
  // slot area = whatever follows the fixed part; a job whose slot does not fit sets status bit 1
  // (needed: sum over selected chunks of align16(len+5), twice where a base exists)
  const uint64_t avail = ws_bytes - w.fixed_bytes;
  HMSE_HIP(hipMemsetAsync(w.counter, 0, sizeof(unsigned long long), stream));
  const uint32_t blocks = (uint32_t)((n_sel + 255) / 256);
  slot_size_kernel<<<dim3(blocks), dim3(256), 0, stream>>>(cuts, chunk_ids, base, n_sel, w.slot_off);
  HMSE_LAUNCH_CHECK();
  if (exclusive_scan_u64(w.slot_off, n_sel, w.slot_off, w.bsum, w.slot_total, stream) != HMSE_OK) return HMSE_EHIP;
  Args a;
  a.data = data; a.n = n; a.cuts = cuts; a.chunk_ids = chunk_ids; a.base = base; a.n_sel = n_sel;
  a.depth = hmse_deflate_depth(cfg);
  a.slot_off = w.slot_off; a.slots = w.slots; a.len_full = w.len_full; a.len_delta = w.len_delta;
  a.slot_cap = avail; a.status = status;
  a.scratch = w.scratch; a.scratch_stride = hmse_align_up(sizeof(Scratch), 256); a.counter = w.counter;
  static bool attr_set = false;
  if (!attr_set) {
    HMSE_HIP(hipFuncSetAttribute((const void*)l1_deflate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Shared)));
    attr_set = true;
  }
  uint64_t grid = 2 * n_sel < (uint64_t)N_WG ? 2 * n_sel : (uint64_t)N_WG;
  PROF_BEGIN(HMSE_STAGE_L1_DEFLATE, stream);
  l1_deflate_kernel<<<dim3((uint32_t)grid), dim3(NT), sizeof(Shared), stream>>>(a);
  PROF_END(HMSE_STAGE_L1_DEFLATE, stream);
  HMSE_LAUNCH_CHECK();
  decide_kernel<<<dim3(blocks), dim3(256), 0, stream>>>(cuts, chunk_ids, base, n_sel, w.len_full, w.len_delta,
                                                        cfg->delta_max_ratio_pct, w.final_len, kind, status);
  HMSE_LAUNCH_CHECK();
  if (exclusive_scan_u64(w.final_len, n_sel, out_off, w.bsum, out_off + n_sel, stream) != HMSE_OK) return HMSE_EHIP;
  gather_kernel<<<dim3((uint32_t)n_sel), dim3(256), 0, stream>>>(cuts, chunk_ids, n_sel, w.slot_off, w.slots, kind, out_off, out,
                                                                out_cap, status);
  HMSE_LAUNCH_CHECK();
  return HMSE_OK;
}
