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

// MEMO TABLE (round 2).  The 128 hashes of a shingle are a pure function of its 4 bytes, text re-uses its 4-grams
// endlessly (256 MiB of the wiki-synth corpus hold 265 k distinct ones, every 8 KiB chunk ~4.4 k of them), and a shingle
// can only lower sig[s] if its hash for seed s is small.  So the first workgroup that meets a shingle computes its 128
// hashes (the tail loop below, as before) and records in a global table which seeds fall below TAU = 2^23 — none for
// 78 % of all shingles, exactly one for 19.5 %, more for 2.6 % ("uncacheable": always computed in full) — as ONE 8-byte
// entry {R, payload}, written by one 64-bit CAS.  Later workgroups look every distinct shingle up (one 8-byte gather;
// 270 G gathers/s while the table stays L2-resident, tools/ubench/gmem_gather.hip) and run the tail loop only over the
// misses.  EXACT: a seed whose minimum over the looked-up and the computed shingles is below TAU has its true minimum
// (every hash below TAU is in the table or was computed); the few seeds that end at or above TAU (0.02 % for a chunk of
// 4.4 k shingles, 2 % for one of 2 k; with TAU = 2^22 this re-evaluation was a quarter of the kernel) are re-evaluated over all distinct shingles of the chunk.  A stale or missing entry is a miss, so nothing depends on when another
// XCD's L2 shows an insert (measured inside ONE cold launch: 95 % of the lookups hit; whether the table has filled up
// is asked once per pass with an L2-bypassing load, because every insert costs two ballots and a global CAS).
// hmse_l4_minhash clears the table per call.  Capacity 2^18 slots (2 MiB: the table has to stay L2-resident beside the streaming corpus — 2^17: 24.5 ms, 2^18: 21.7,
// 2^19: 28.4, 2^20: 27.3 ms per 2 GiB), filled to one half with the first 131 k distinct shingles met (the frequent ones
// come early), then no more inserts: data without repeating 4-grams (random
// bytes) costs what it cost before plus the lookups of the first MH_SAMPLE shingles per wavefront, after which the
// wavefront stops looking.
constexpr uint32_t MH_TAU = 1u << 23;      // payload: value [22:0], seed [29:23]
constexpr uint32_t MH_P_NONE = 1u << 30, MH_P_UNC = 1u << 31;
#ifndef HMSE_MH_MEMO_BITS
#define HMSE_MH_MEMO_BITS 18
#endif
constexpr int MH_MEMO_BITS = HMSE_MH_MEMO_BITS;
#ifndef HMSE_MH_CAP
#define HMSE_MH_CAP (1u << (MH_MEMO_BITS - 1))
#endif
constexpr uint64_t MH_MEMO_EMPTY = ~0ull;
constexpr uint32_t MH_SAMPLE = 256;     // lookups after which a wavefront with < 1/4 hits stops looking (per pass)
struct MhMemo { unsigned long long* tab; uint32_t* count; uint32_t bits; uint32_t cap; uint32_t probe; };   // probe: diagnostic build only

constexpr int MH_TBITS = 14;
constexpr int MH_SLOTS = 1 << MH_TBITS;   // 16384 x 4 B = 64 KiB
constexpr int MH_SUB = MH_SLOTS * 3 / 4;  // shingles per pass: load factor <= 0.75 even if all are distinct (text: ~0.4);
                                          // 12288 covers a typical 8-12 KiB chunk in ONE pass, so its whole shingle set is de-duplicated
constexpr uint32_t MH_EMPTY = 0xFFFFFFFFu;
// workgroups of a launch, each takes every MH_GRID-th chunk.  Measured at 10 GB (940 k chunks): 512 (= the resident slots: a static share each)
// 46.3 ms, 1 024 44.9, 2 048 43.8, 4 096 43.3, 16 384 .. 65 536 42.8, 262 144 43.7, one workgroup per chunk 45.8 — few workgroups balance badly,
// many pay their launches.
constexpr uint64_t MH_GRID = 16384;

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

#ifdef HMSE_DIAG
__device__ unsigned long long g_mh_stamps[16];   // diagnostic build: thread 0's clocks per phase (tools/minhash_stamps.py)
#define MH_STAMP(i) do { if (threadIdx.x == 0) { const unsigned long long now__ = clock64(); mh_acc[i] += now__ - mh_last; mh_last = now__; } } while (0)
#else
#define MH_STAMP(i) do { } while (0)
#endif

// (launch bounds: 64 VGPRs, so that two 1024-thread workgroups share a CU — at 72 registers, which two experiments of round 4 reached, one
// workgroup per CU runs and the kernel takes 15.9 instead of 11.2 ms at 2 GB)
template <int NT, int V>
__global__ __launch_bounds__(NT, NT == 1024 ? 8 : 4) void l4_minhash_kernel(const uint8_t* __restrict__ data, uint64_t n,
                                                        const uint64_t* __restrict__ cuts,
                                                        const uint64_t* __restrict__ chunk_ids, uint64_t n_sel,
                                                        uint32_t seed_base, uint32_t* __restrict__ sig, const uint64_t* __restrict__ st,
                                                        MhMemo memo, uint64_t sel0) {
  __shared__ __attribute__((aligned(16))) uint32_t s_tab[MH_SLOTS];
  __shared__ uint32_t s_sig[NT / 64][128];
  __shared__ uint32_t s_min[128], s_pass[128], s_glob[128], s_unres[128];
  __shared__ uint32_t s_flag, s_nun, s_any, s_ins;
  if (st) { chunk_ids += st[SB_U_OLD]; sig += 128 * st[SB_U_OLD]; n_sel = st[SB_U_NEW]; }   // captured chain: this batch's stored chunks
#ifdef HMSE_DIAG
  unsigned long long mh_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, mh_last = clock64();
#endif
  const uint32_t t = threadIdx.x, lane = lane_id(), w = t >> 6;
  // A workgroup takes chunks blockIdx.x, blockIdx.x + gridDim.x, ... (round 4; one workgroup per chunk before: 45.8 -> 42.8 ms at 10 GB).  The
  // captured chain has to size its grid for the most chunks a batch CAN hold — five times what a 1 GiB batch of text does — and launched
  // that many workgroups per batch, most of them empty.
  for (uint64_t sel = sel0 + blockIdx.x; sel < n_sel; sel += gridDim.x) {
  const uint64_t c = chunk_ids ? chunk_ids[sel] : sel;
  const uint64_t start = cuts[c];
  const uint64_t len = cuts[c + 1] - start;
  const uint64_t nsh = len >= 4 ? len - 3 : 0;
  const uint32_t sr0 = rotl32(seed_base + lane, 13), sr1 = rotl32(seed_base + lane + 64, 13);
  constexpr int PART = MH_SLOTS / (NT / 64);  // each wavefront compacts and streams its own part of the table
  uint32_t* q = s_tab + w * PART;
  if (t < 128) { s_glob[t] = 0xFFFFFFFFu; s_min[t] = 0xFFFFFFFFu; }
  if (t == 0) { s_nun = 0; s_any = 0; }
  MH_STAMP(0);   // prologue: the chunk's metadata

  for (uint64_t sub0 = 0; sub0 < nsh; sub0 += MH_SUB) {
    const uint32_t cnt = (uint32_t)((nsh - sub0) < (uint64_t)MH_SUB ? (nsh - sub0) : (uint64_t)MH_SUB);
    // clear the set
    for (uint32_t i = t; i < MH_SLOTS / 4; i += NT) ((uint4*)s_tab)[i] = make_uint4(MH_EMPTY, MH_EMPTY, MH_EMPTY, MH_EMPTY);
    if (t == 0) {
      s_flag = 0;
      // room for new entries?  Asked once per pass with a load that bypasses this XCD's L2: in a launch that starts cold, the
      // workgroups that start later must see that the table has filled up (every insert is two ballots and a global CAS)
      s_ins = memo.tab && __hip_atomic_load(memo.count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < memo.cap;
    }
    __syncthreads();
    MH_STAMP(1);   // clear + "room in the table?"
    const bool ins = s_ins != 0;
    // insert R of every shingle of this pass
    // Four consecutive shingles per lane from three ALIGNED, coalesced dword loads (round 4): a 4-byte load at a byte-granular address per
    // lane — what this loop did, one per shingle — is handled lane by lane by the memory pipeline; the aligned form is a quarter of the load
    // instructions, each a plain coalesced 256-byte read.  (Shingle k of the chunk = bytes [k, k + 4): the same values, only fetched differently.)
    auto set_insert = [&](uint32_t x) {
      const uint32_t R = murmur_R(x);
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
    };
    {
      const uint64_t b0 = start + sub0;                      // byte offset of this pass's first shingle
      const uint32_t sh = (uint32_t)(b0 & 3u);
      const uint8_t* const abase = data + (b0 - sh);         // 4-byte aligned (data comes from the allocator: 256-byte aligned; offsets are bytes)
      const bool base_aligned = (((uintptr_t)data) & 3u) == 0;
      for (uint32_t p4 = t * 4u; p4 < (V == 3 ? 0u : cnt); p4 += NT * 4u) {
        const uint64_t a = (b0 - sh) + p4;
        if (base_aligned && a + 12 <= n) {
          const uint32_t* w = (const uint32_t*)(abase + p4);
          const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
          const uint32_t x0 = __builtin_amdgcn_alignbyte(w1, w0, sh), y0 = __builtin_amdgcn_alignbyte(w2, w1, sh);   // bytes [p4, p4 + 4), [p4 + 4, p4 + 8)
          set_insert(x0);
          if (p4 + 1u < cnt) set_insert(__builtin_amdgcn_alignbyte(y0, x0, 1u));
          if (p4 + 2u < cnt) set_insert(__builtin_amdgcn_alignbyte(y0, x0, 2u));
          if (p4 + 3u < cnt) set_insert(__builtin_amdgcn_alignbyte(y0, x0, 3u));
        } else {
          const uint8_t* src = data + b0;
          for (uint32_t k = 0; k < 4u && p4 + k < cnt; k++) set_insert(load_u32_unaligned(src + p4 + k));
        }
      }
    }
    __syncthreads();
    MH_STAMP(2);   // set inserts
    // in-place compaction of this wave's part (write index never passes the read index)
    uint32_t wr = 0;
    static_assert(PART % 256 == 0, "the compaction reads four 64-entry rows per round trip");
    for (uint32_t i = 0; i < (uint32_t)PART; i += 256) {   // (four independent reads in flight, then the four appends: all of a group's reads precede its writes)
      uint32_t v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) v[u] = q[i + 64 * u + lane];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const bool valid = v[u] != MH_EMPTY;
        const uint64_t m = __ballot(valid);
        if (valid) q[wr + mbcnt64(m)] = v[u];
        wr += (uint32_t)__builtin_popcountll(m);
      }
    }
    MH_STAMP(3);   // compaction
    // ---- memo lookups: every distinct shingle of this wavefront's part; what the table cannot answer goes to the top of
    // the part (the todo list), which the tail loop below then streams instead of the whole part
    uint32_t m0 = 0xFFFFFFFFu, m1 = 0xFFFFFFFFu;   // this pass, this wavefront: minima over the computed shingles
    const uint32_t* tp = q;
    uint32_t nt = wr;
    if (V == 1 && memo.tab && wr * 2 <= (uint32_t)PART) {
      uint32_t looked = 0, hits = 0, i0 = 0;
      nt = 0;
      const uint32_t mask = (1u << memo.bits) - 1u;
      for (; i0 < wr; i0 += 256) {          // four entries per lane: their first probes are in flight together
        uint32_t R[4]; unsigned long long e[4]; uint32_t sl[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const uint32_t i = i0 + 64 * u + lane;
          R[u] = i < wr ? q[i] : 0u;
          sl[u] = (R[u] * 0x9E3779B1u) >> (32 - memo.bits);
#ifdef HMSE_DIAG
          if (memo.probe & 2u) e[u] = ((unsigned long long)MH_P_NONE << 32) | R[u]; else   // timing probe: no table reads, wrong signatures
#endif
          e[u] = i < wr ? memo.tab[sl[u]] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const uint32_t i = i0 + 64 * u + lane;
          const bool valid = i < wr;
          bool todo = false;
          if (valid) {
            unsigned long long ee = e[u];
            if ((uint32_t)ee != R[u] && ee != MH_MEMO_EMPTY) {
              uint32_t s2 = (sl[u] + 1) & mask; ee = memo.tab[s2];
              if ((uint32_t)ee != R[u] && ee != MH_MEMO_EMPTY) { s2 = (s2 + 1) & mask; ee = memo.tab[s2]; }
            }
            if ((uint32_t)ee == R[u]) {
              const uint32_t pl = (uint32_t)(ee >> 32);
              if (pl & MH_P_UNC) todo = true;
              else if (!(pl & MH_P_NONE)) atomicMin(&s_min[(pl >> 23) & 127u], pl & (MH_TAU - 1u));
            } else todo = true;
          }
          const uint64_t mt = __ballot(todo);
          if (todo) q[PART - 1 - (nt + (uint32_t)__builtin_popcountll(mt & lanemask_lt()))] = R[u];
          nt += (uint32_t)__builtin_popcountll(mt);
          looked += (uint32_t)__builtin_popcountll(__ballot(valid));
        }
        hits = looked - nt;
        if (looked >= MH_SAMPLE && hits * 4 < looked) { i0 += 256; break; }   // a cold or useless table: stop looking
      }
      for (; i0 < wr; i0 += 64) {      // (the rest of the part after giving up: all todo)
        const uint32_t i = i0 + lane;
        if (i < wr) q[PART - 1 - (nt + (i - i0))] = q[i];
        nt += (wr - i0) < 64u ? (wr - i0) : 64u;
      }
      tp = q + PART - nt;
      if (lane == 0) s_any = 1;
#ifdef HMSE_DIAG
      if (lane == 0 && (memo.probe & 4u)) { atomicAdd(memo.count + 1, looked); atomicAdd(memo.count + 2, nt); atomicAdd(memo.count + 4, wr); }   // tools/minhash_memo_stats.py
#endif
    }
    MH_STAMP(4);   // look-ups
    // one computed shingle -> the table: which of its 128 hashes fall below TAU
    auto memo_insert = [&](uint32_t R, uint32_t a0, uint32_t a1) {
      const uint64_t b0 = __ballot(a0 < MH_TAU), b1 = __ballot(a1 < MH_TAU);
      const uint32_t cnt = (uint32_t)__builtin_popcountll(b0) + (uint32_t)__builtin_popcountll(b1);
      uint32_t pl = MH_P_NONE;
      if (cnt >= 2) pl = MH_P_UNC;
      else if (cnt == 1) {
        const uint32_t sl = b0 ? (uint32_t)__builtin_ctzll(b0) : (uint32_t)__builtin_ctzll(b1);
        const uint32_t val = (uint32_t)__builtin_amdgcn_readlane((int)(b0 ? a0 : a1), (int)sl);
        pl = ((b0 ? sl : sl + 64u) << 23) | val;
      }
      if (lane == 0) {
        const unsigned long long ent = ((unsigned long long)pl << 32) | R;
        const uint32_t mask = (1u << memo.bits) - 1u;
        uint32_t sl = (R * 0x9E3779B1u) >> (32 - memo.bits);
        for (int pr = 0; pr < 3; pr++) {
          const unsigned long long old = atomicCAS(&memo.tab[sl], MH_MEMO_EMPTY, ent);
          if (old == MH_MEMO_EMPTY) { atomicAdd(memo.count, 1u); break; }
          if ((uint32_t)old == R) break;
          sl = (sl + 1) & mask;
        }
      }
      __builtin_amdgcn_wave_barrier();
    };
    // stream the todo list (without a table: the whole part) through the 2 seeds of this lane
    // (Round 4, measured and rejected, both at 64 registers: POOLING the wavefronts' todo lists so that every wavefront streams an equal share —
    // the barrier behind this loop waits 7-15 % of thread 0's clocks for the fullest part, tools/minhash_stamps.py — needs a barrier in front of
    // the loop, which stops the early wavefronts' tails from running beside the late wavefronts' gathers: 11.4 -> 11.8 ms at 2 GB; the chunk's
    // byte loads issued a group ahead, the first one in front of the clear: 11.2 against 11.1 — the CU's other workgroup already hides them.)
    if (V == 2) nt = 0;
    uint32_t hd = (4u - ((uint32_t)(tp - q) & 3u)) & 3u;      // entries in front of the first 16-byte boundary
    if (hd > nt) hd = nt;
    for (uint32_t i = 0; i < hd; i++) {
      const uint32_t R = tp[i];
      const uint32_t a0 = murmur_tail<V>(R, sr0), a1 = murmur_tail<V>(R, sr1);
      m0 = min(m0, a0); m1 = min(m1, a1);
      if (ins) memo_insert(R, a0, a1);
    }
    const uint32_t n4 = hd + ((nt - hd) & ~3u);
    for (uint32_t i = hd; i < n4; i += 4) {
      const uint4 kv = *(const uint4*)(tp + i);
      const uint32_t a0 = murmur_tail<V>(kv.x, sr0), a1 = murmur_tail<V>(kv.x, sr1);
      const uint32_t b0 = murmur_tail<V>(kv.y, sr0), b1 = murmur_tail<V>(kv.y, sr1);
      const uint32_t c0 = murmur_tail<V>(kv.z, sr0), c1 = murmur_tail<V>(kv.z, sr1);
      const uint32_t d0 = murmur_tail<V>(kv.w, sr0), d1 = murmur_tail<V>(kv.w, sr1);
      m0 = min3u(min3u(m0, a0, b0), c0, d0);
      m1 = min3u(min3u(m1, a1, b1), c1, d1);
      if (ins) { memo_insert(kv.x, a0, a1); memo_insert(kv.y, b0, b1); memo_insert(kv.z, c0, c1); memo_insert(kv.w, d0, d1); }
    }
    for (uint32_t i = n4; i < nt; i++) {
      const uint32_t R = tp[i];
      const uint32_t a0 = murmur_tail<V>(R, sr0), a1 = murmur_tail<V>(R, sr1);
      m0 = min(m0, a0); m1 = min(m1, a1);
      if (ins) memo_insert(R, a0, a1);
    }
    if (w == 0 && s_flag) {  // the one R value that collides with the empty marker
      m0 = min(m0, murmur_tail<V>(MH_EMPTY, sr0));
      m1 = min(m1, murmur_tail<V>(MH_EMPTY, sr1));
    }
    // ---- this pass's minima: computed (per wavefront) and looked up (s_min); seeds still at or above TAU where a table
    // was consulted are re-evaluated over every distinct shingle of the pass
    MH_STAMP(5);   // tail: the shingles the table could not answer
    s_sig[w][lane] = m0;
    s_sig[w][lane + 64] = m1;
    __syncthreads();
    MH_STAMP(6);   // barrier behind the tail (the slowest wavefront)
    if (t < 128) {
      uint32_t v = s_min[t];
#pragma unroll
      for (int i = 0; i < NT / 64; i++) v = min(v, s_sig[i][t]);
      s_pass[t] = v;
      if (s_any && v >= MH_TAU) s_unres[atomicAdd(&s_nun, 1u)] = t;
    }
    __syncthreads();
#ifdef HMSE_DIAG
    const uint32_t nun = (memo.probe & 1u) ? 0u : s_nun;   // timing probe: no re-evaluation, wrong signatures
#else
    const uint32_t nun = s_nun;
#endif
#ifdef HMSE_DIAG
    if (t == 0 && memo.tab && (memo.probe & 4u)) { atomicAdd(memo.count + 3, nun); atomicAdd(memo.count + 5, 1u); }
#endif
    for (uint32_t j = 0; j < nun; j++) {
      const uint32_t sd = s_unres[j];
      const uint32_t sr = rotl32(seed_base + sd, 13);
      uint32_t lo = 0xFFFFFFFFu;
      for (uint32_t i = lane; i < wr; i += 64) lo = min(lo, murmur_tail<V>(q[i], sr));
      if (w == 0 && s_flag) lo = min(lo, murmur_tail<V>(MH_EMPTY, sr));
      lo = ~wave_max(~lo);   // minimum over the wavefront on the DPP path
      if (lane == 0) atomicMin(&s_pass[sd], lo);
    }
    __syncthreads();
    if (t < 128) { s_glob[t] = min(s_glob[t], s_pass[t]); s_min[t] = 0xFFFFFFFFu; }
    if (t == 0) { s_nun = 0; s_any = 0; }
    __syncthreads();
    MH_STAMP(7);   // combine + re-evaluation
  }
  if (t < 128) sig[sel * 128 + t] = s_glob[t];
  __syncthreads();   // (the next chunk's first writes to the shared state come behind every read of this one's)
#ifdef HMSE_DIAG
  MH_STAMP(8);
  if (threadIdx.x == 0) atomicAdd(&g_mh_stamps[15], 1ull);
#endif
  }
#ifdef HMSE_DIAG
  if (threadIdx.x == 0) for (int i = 0; i < 9; i++) atomicAdd(&g_mh_stamps[i], mh_acc[i]);
#endif
}

#ifdef HMSE_DIAG
extern "C" int hmse_debug_minhash_stamps(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_mh_stamps), sizeof(g_mh_stamps)) != hipSuccess) return HMSE_EHIP;
  if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_mh_stamps), z, sizeof z) != hipSuccess) return HMSE_EHIP; }
  return HMSE_OK;
}
#endif

// workspace: [0, 256) header (entry count), then the memo table
size_t hmse_l4_minhash_workspace_bytes_impl(uint64_t) { return 256 + ((size_t)8 << MH_MEMO_BITS); }

extern "C" int hmse_l4_minhash(const uint8_t* data, uint64_t n, const uint64_t* cuts, const uint64_t* chunk_ids,
                               uint64_t n_sel, const hmse_cfg* cfg, uint32_t* sig, void* ws, size_t ws_bytes, void* stream_) {
  if (hmse_cfg_validate_impl(cfg) != 0) return HMSE_EINVAL;
  if (n_sel == 0) return HMSE_OK;
  if (!data || !cuts || !sig) return HMSE_EINVAL;
  if (n_sel > 0x7FFFFFFFull) return HMSE_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  (void)hipGetLastError();  // drop stale errors of earlier runtime calls made by the host process
  // the memo table lives in the caller's workspace and is cleared per call (a call's result and cost do not depend on
  // what ran before); a workspace of the old size (256 bytes) simply runs without one
  MhMemo memo{nullptr, nullptr, MH_MEMO_BITS, HMSE_MH_CAP, 0};
#ifdef HMSE_DIAG
  if (getenv("HMSE_MH_PROBE")) memo.probe = (uint32_t)atoi(getenv("HMSE_MH_PROBE"));   // 1 no re-evaluation, 2 no table reads, 4 counters
#endif
  if (ws && ws_bytes >= hmse_l4_minhash_workspace_bytes_impl(n_sel)) {
    memo.count = (uint32_t*)ws;
    memo.tab = (unsigned long long*)((uint8_t*)ws + 256);
    HMSE_FILL(ws, 0, 256, stream);
    HMSE_FILL(memo.tab, 0xFF, (size_t)8 << MH_MEMO_BITS, stream);
  }
  PROF_BEGIN(HMSE_STAGE_L4_MINHASH, stream);
#ifdef HMSE_DIAG
  // diagnostic build only (libhmse_hip_diag.so, tools/minhash_variants.py): threads per workgroup x instruction variant;
  // variants 6 and 7 are timing probes that return WRONG signatures, which is why none of this is in the product library
  static const int variant = getenv("HMSE_MH_VARIANT") ? atoi(getenv("HMSE_MH_VARIANT")) : MH_DEFAULT_VARIANT;
  const dim3 grid((uint32_t)n_sel);
  const MhMemo none{nullptr, nullptr, MH_MEMO_BITS, 0, 0};
  switch (variant) {
    case 0: l4_minhash_kernel<256, 0><<<grid, dim3(256), 0, stream>>>(data, n, cuts, chunk_ids, n_sel, cfg->seed_base, sig, nullptr, none, 0); PROF_END(HMSE_STAGE_L4_MINHASH, stream); HMSE_LAUNCH_CHECK(); return HMSE_OK;
    case 1: l4_minhash_kernel<256, 1><<<grid, dim3(256), 0, stream>>>(data, n, cuts, chunk_ids, n_sel, cfg->seed_base, sig, nullptr, none, 0); PROF_END(HMSE_STAGE_L4_MINHASH, stream); HMSE_LAUNCH_CHECK(); return HMSE_OK;
    case 2: l4_minhash_kernel<512, 0><<<grid, dim3(512), 0, stream>>>(data, n, cuts, chunk_ids, n_sel, cfg->seed_base, sig, nullptr, none, 0); PROF_END(HMSE_STAGE_L4_MINHASH, stream); HMSE_LAUNCH_CHECK(); return HMSE_OK;
    case 3: l4_minhash_kernel<512, 1><<<grid, dim3(512), 0, stream>>>(data, n, cuts, chunk_ids, n_sel, cfg->seed_base, sig, nullptr, none, 0); PROF_END(HMSE_STAGE_L4_MINHASH, stream); HMSE_LAUNCH_CHECK(); return HMSE_OK;
    case 4: l4_minhash_kernel<1024, 0><<<grid, dim3(1024), 0, stream>>>(data, n, cuts, chunk_ids, n_sel, cfg->seed_base, sig, nullptr, none, 0); PROF_END(HMSE_STAGE_L4_MINHASH, stream); HMSE_LAUNCH_CHECK(); return HMSE_OK;
    case 6: l4_minhash_kernel<1024, 2><<<grid, dim3(1024), 0, stream>>>(data, n, cuts, chunk_ids, n_sel, cfg->seed_base, sig, nullptr, none, 0); PROF_END(HMSE_STAGE_L4_MINHASH, stream); HMSE_LAUNCH_CHECK(); return HMSE_OK;  // timing probe: no tail
    case 7: l4_minhash_kernel<1024, 3><<<grid, dim3(1024), 0, stream>>>(data, n, cuts, chunk_ids, n_sel, cfg->seed_base, sig, nullptr, none, 0); PROF_END(HMSE_STAGE_L4_MINHASH, stream); HMSE_LAUNCH_CHECK(); return HMSE_OK;  // timing probe: no inserts
    case 8: memo = none; break;   // the product kernel without its memo table
    default: break;
  }
#endif
  l4_minhash_kernel<1024, 1><<<dim3((uint32_t)(n_sel < MH_GRID ? n_sel : MH_GRID)), dim3(1024), 0, stream>>>(data, n, cuts, chunk_ids, n_sel, cfg->seed_base, sig, nullptr, memo, 0);
  PROF_END(HMSE_STAGE_L4_MINHASH, stream);
  HMSE_LAUNCH_CHECK();
  return HMSE_OK;
}

// The captured chain's memo table PERSISTS across the batches of a stream (it caches a pure function of (shingle, seed_base):
// nothing in it depends on the data seen so far).  Cleared per batch, every 1 GiB batch paid the table's warm-up again: MinHash
// 8.2 ms per GiB in a stream against 5.3 in a 10 GB one-shot call.  hmse_stream_workspace_init() clears it ONCE and tags the
// header; a chain that finds no tag (workspace never initialised, or initialised for another seed_base) sets sticky status bit 4.
constexpr unsigned long long MH_MEMO_MAGIC = 0x484D53454D454D4Full;   // "HMSEMEMO"
__global__ void mh_memo_tag_kernel(unsigned long long* hdr, unsigned long long tag) { if (threadIdx.x == 0 && blockIdx.x == 0) hdr[8] = tag; }
__global__ void mh_memo_check_kernel(const unsigned long long* hdr, unsigned long long tag, uint64_t* st) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && hdr[8] != tag) atomicOr((unsigned long long*)&st[SB_STATUS], 16ull);
}
int hmse_l4_minhash_memo_init(void* ws, size_t ws_bytes, const hmse_cfg* cfg, hipStream_t stream) {
  if (!ws || ws_bytes < hmse_l4_minhash_workspace_bytes_impl(1)) return HMSE_ENOSPC;
  HMSE_FILL(ws, 0, 256, stream);
  HMSE_FILL((uint8_t*)ws + 256, 0xFF, (size_t)8 << MH_MEMO_BITS, stream);
  mh_memo_tag_kernel<<<dim3(1), dim3(64), 0, stream>>>((unsigned long long*)ws, MH_MEMO_MAGIC ^ cfg->seed_base);
  HMSE_LAUNCH_CHECK();
  return HMSE_OK;
}

// captured chain: one workgroup per POSSIBLE stored chunk of the batch; those beyond the device-side count leave at once
int hmse_l4_minhash_dyn(const uint8_t* data, uint64_t n_cap, const uint64_t* cuts_all, const uint64_t* uniq_all, uint32_t* sig_all,
                        const uint64_t* st, uint64_t cap_chunks, const hmse_cfg* cfg, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (!data || !cuts_all || !uniq_all || !sig_all || !st || cap_chunks == 0 || cap_chunks > 0x7FFFFFFFull) return HMSE_EINVAL;
  MhMemo memo{nullptr, nullptr, MH_MEMO_BITS, HMSE_MH_CAP, 0};
  if (ws && ws_bytes >= hmse_l4_minhash_workspace_bytes_impl(cap_chunks)) {   // the stream's memo table (hmse_stream_workspace_init)
    memo.count = (uint32_t*)ws;
    memo.tab = (unsigned long long*)((uint8_t*)ws + 256);
    mh_memo_check_kernel<<<dim3(1), dim3(64), 0, stream>>>((const unsigned long long*)ws, MH_MEMO_MAGIC ^ cfg->seed_base, (uint64_t*)st);
  }
  l4_minhash_kernel<1024, 1><<<dim3((uint32_t)(cap_chunks < MH_GRID ? cap_chunks : MH_GRID)), dim3(1024), 0, stream>>>(data, n_cap, cuts_all, uniq_all, 0, cfg->seed_base, sig_all, st, memo, 0);
  HMSE_LAUNCH_CHECK();
  return HMSE_OK;
}
