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
//   2. every chunk position finds its longest match in parallel (lanes are small state machines that pull the next
//      sorted rank when they finish one);
//   3. the one-step-lazy parse p -> next(p) is a functional graph: every wavefront walks its own segment
//      speculatively (a chain of v_readlane inside 64-position windows), wave 0 stitches the segments where
//      the chains meet;
//   4. tokens compacted in parse order + symbol histograms (LDS atomics) -> a job record in HBM     [match kernel]
//   5. length-limited Huffman (sorted two-queue merge + Kraft repair), closed-form code-length RLE, smallest of
//      stored/fixed/dynamic; token bit offsets by prefix scans, bits OR-ed into an LDS image, coalesced
//      copy-out                                                                                      [encode kernel]
// Match kernel: one 512/1024-thread workgroup per job (chunk, or chunk + dictionary), size classes by window
// length; encode kernel: one 256-thread workgroup per job, six per CU.  Persistent workgroups pull jobs from
// per-class lists.
#include "common.h"
#include <stdio.h>
#include <stdlib.h>
#include <mutex>

namespace dfl {

constexpr int HB = 12;
constexpr int NBK = 1 << HB;
constexpr uint32_t MINM = 4, MAXM = 258, WMAX = 32768;

__device__ __forceinline__ uint32_t hash4(uint32_t x) { return (x * 0x9E3779B1u) >> (32 - HB); }
__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

// Window bytes at an ARBITRARY byte offset, fetched with naturally aligned dword reads and shifted into place in registers
// (v_alignbyte_b32).  Measured on gfx950 (tools/ubench/lds_unaligned.hip, profiles/r2/lds_unaligned_gfx950.txt): an LDS read
// whose per-lane addresses are off the instruction's natural alignment — ds_read_b32 included — costs ~64 LDS clocks per
// wave-instruction (one per lane) against 6.7 for an aligned random ds_read_b32; the matcher compares window bytes at
// arbitrary offsets all the time, so every such read goes through these helpers.  `W` is 16-byte aligned.
__device__ __forceinline__ uint32_t alignb(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }
__device__ __forceinline__ uint32_t ld32a(const uint8_t* W, uint32_t off) {
  const uint32_t* w = (const uint32_t*)(W + (off & ~3u));
  return alignb(w[1], w[0], off);
}
__device__ __forceinline__ void ld64a(const uint8_t* W, uint32_t off, uint32_t& x0, uint32_t& x1) {
  const uint32_t* w = (const uint32_t*)(W + (off & ~3u));
  const uint32_t a = w[0], b = w[1], c = w[2];
  x0 = alignb(b, a, off); x1 = alignb(c, b, off);
}
// The sort's key of position `off`: the hash product of its four bytes in bits 16..31 (bucket = bits 20..31, filter bits = 16..19) and the
// low nibble of its byte 4 in bits 0..3 — all from the two aligned dwords the hash needs anyway (byte 4 lies in the second one).
__device__ __forceinline__ uint32_t sort_key(const uint8_t* W, uint32_t off) {
  const uint32_t* w = (const uint32_t*)(W + (off & ~3u));
  const uint32_t a = w[0], b = w[1];
  return ((alignb(b, a, off) * 0x9E3779B1u) & 0xFFFF0000u) | ((b >> ((off & 3u) * 8u)) & 0x0Fu);
}
// N * 8 bytes at W + off as 64-bit pieces
template <int N>
__device__ __forceinline__ void ldNa(const uint8_t* W, uint32_t off, uint64_t (&x)[N]) {
  const uint32_t* w = (const uint32_t*)(W + (off & ~3u));
  uint32_t d[2 * N + 1];
#pragma unroll
  for (int u = 0; u < 2 * N + 1; u++) d[u] = w[u];
#pragma unroll
  for (int u = 0; u < N; u++) x[u] = ((uint64_t)alignb(d[2 * u + 2], d[2 * u + 1], off) << 32) | alignb(d[2 * u + 1], d[2 * u], off);
}

// bucket cursors: two 16-bit counters per dword (values <= 65533), updated with 32-bit LDS atomics
__device__ __forceinline__ uint32_t cur_get(const uint32_t* c, uint32_t h) { return (c[h >> 1] >> ((h & 1u) * 16u)) & 0xFFFFu; }
__device__ __forceinline__ uint32_t cur_inc(uint32_t* c, uint32_t h) {
  const uint32_t sh = (h & 1u) * 16u;
  return (atomicAdd(&c[h >> 1], 1u << sh) >> sh) & 0xFFFFu;
}

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

// A value that IS the same in every lane, told to the compiler (SGPRs): control flow that depends only on such values becomes
// scalar branches — no EXEC masking, so a cross-lane read inside it (readfirstlane, bpermute from lane 0) always finds its source
// lane active.  (Round 3: the dictionary jobs' work queue was wave-uniform in fact but divergent in the compiler's analysis — the
// chunk length it compared against came from a vector load — and the structurised loop it produced never finished on the first
// shard with more class-S dictionary jobs than resident workgroups; profiles/r3/r3_deflate_dict_queue_hang_isa.txt.)
__device__ __forceinline__ uint32_t uni32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uni64(uint64_t v) { return ((uint64_t)uni32((uint32_t)(v >> 32)) << 32) | uni32((uint32_t)v); }

// order LDS traffic between the lanes of one wavefront (no instruction is emitted for the barrier itself)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

template <int N>
struct HuffScratchT {
  uint32_t key[N];         // compacted (freq << 9 | sym)
  uint32_t sorted[N];      // ascending
  uint32_t w[2 * N];       // node weights: leaves then internals
  uint16_t parent[2 * N];
  uint32_t cnt[32];        // codes per length
};
using HuffL = HuffScratchT<288>;
using HuffD = HuffScratchT<32>;

// small per-workgroup state of the match kernel (always in LDS; the encode kernel has its own: EncSmall)
template <int NT, bool WQ>
struct Small {
  uint32_t lf[288], df[32];   // literal / length and distance histograms of the job's tokens
  uint32_t red[NT / 64 + 1];
  uint32_t qhead;
  uint32_t ocnt[4];   // sorted ranks per chain-length class (the matcher's hand-out order)
  struct { uint32_t ji, job, L, Dl; uint64_t c, cstart, rec_at, dstart; } nx;   // the NEXT job's metadata, fetched while this one runs
  uint32_t pexit[NT / 64], pexit2[NT / 64], pconv[NT / 64];
  uint8_t wtab[WQ ? NT : 64];   // class B's dictionary jobs (no hand-out list): per wavefront, lane that holds the r-th pending work rank of the wave's window
};

constexpr int align16(int v) { return (v + 15) & ~15; }

// Size classes of the match kernel by window T = dictionary + chunk (the table behind `struct Layout` says what each keeps in LDS).  The caps of
// the two-per-CU classes are what 80 KiB of LDS hold (static_assert in Layout): FastCDC's chunk sizes peak just above the 8 KiB average
// (18 / 14 / 11 / 9 % of the bytes in 8.0-8.5 / 8.5-9.0 / 9.0-9.5 / 9.5-10.0 KiB chunks), and every class is cheaper per byte than the next.
#ifndef HMSE_TCAP_S
#define HMSE_TCAP_S 10048
#define HMSE_TCAP_S2 13952
#define HMSE_TCAP_SG 17408
#define HMSE_TCAP_SG2 22976
#endif
constexpr int NT_S = 1024, TCAP_S = HMSE_TCAP_S, TCAP_S2 = HMSE_TCAP_S2, TCAP_SG = HMSE_TCAP_SG, TCAP_SG2 = HMSE_TCAP_SG2;
constexpr int NT_M = 1024, TCAP_SG3 = 32768;
#ifndef HMSE_NT_B
#define HMSE_NT_B 1024
#endif
constexpr int NT_B = HMSE_NT_B, TCAP_B = 65536, LCAP_B = 32768;

// LDS carve.  LDSM: every per-position array lives in LDS (size classes T <= TCAP);
// !LDSM (rare big jobs): S / jump / match arrays live in a per-workgroup global scratch.
template <int NT, int TCAP, int LCAP_, bool LDSM, bool MDG = false, bool MLG = false, bool NOK = false>
struct Layout {
  static constexpr int LCAP = LCAP_;
  static constexpr int W_OFF = 0;
  static constexpr int W_SZ = align16(TCAP + 64);
  static constexpr int CUR_OFF = W_OFF + W_SZ;          // packed u16[NBK] bucket cursors; later the Huffman scratch
  static constexpr int CUR_SZ = NBK * 2;               // two 16-bit cursors per dword
  static constexpr int MARK_OFF = CUR_OFF + CUR_SZ;
  static constexpr int MARK_SZ = align16(LCAP / 8 + 16);
  static constexpr int SMALL_OFF = MARK_OFF + MARK_SZ;
  static constexpr int SMALL_SZ = align16((int)sizeof(Small<NT, !LDSM>));
  static constexpr int A_OFF = SMALL_OFF + SMALL_SZ;    // LDSM: u16 S[T] -> u16 jump[L+1] -> out image; !LDSM: out image
  static constexpr int A_SZ = LDSM ? align16(2 * TCAP + 80) : align16(LCAP + 80);
  static constexpr int MD_OFF = A_OFF + A_SZ;           // LDSM: u16 mdist[L]
  static constexpr int MD_SZ = (LDSM && !MDG) ? align16(2 * LCAP) : 0;
  static constexpr int ML_OFF = MD_OFF + MD_SZ;         // LDSM: u8 mlen[L]
  static constexpr int ML_SZ = (LDSM && !MLG) ? align16(LCAP) : 0;
  static constexpr int K_OFF = ML_OFF + ML_SZ;          // LDSM: u8 K[T] = byte 4 of the position at each sorted rank
  static constexpr int K_SZ = (LDSM && !NOK) ? align16(TCAP + 16) : 0;
  static constexpr int TOTAL = K_OFF + K_SZ;
  static_assert(TOTAL <= 160 * 1024, "one workgroup's LDS image must fit the CU's 160 KiB");
  static_assert(!(LDSM && TCAP <= TCAP_SG2) || TOTAL <= 80 * 1024, "classes S .. SG2 run two workgroups per CU");
  static_assert(sizeof(HuffL) + sizeof(HuffD) <= CUR_SZ, "Huffman scratch must fit the cursor table");
};

// ---- wave-level Huffman length construction (one wavefront, mirrors oracle huff_lengths) --------
template <typename HS>
__device__ void huff_lengths_wave(const uint32_t* freq, uint32_t n, uint32_t limit, uint8_t* lens, HS* hs) {
  const uint32_t lane = lane_id();
  // (1) at least two used symbols: the lowest unused indices count as frequency 1
  // (the histogram itself is left untouched: dummies only exist inside the tree construction)
  uint32_t used = 0;
  for (uint32_t b = 0; b < n; b += 64) {
    const uint32_t s = b + lane;
    used += (uint32_t)__builtin_popcountll(__ballot(s < n && freq[s] != 0));
  }
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
  // (4) two-queue merge (leaf wins ties), serial on lane 0.  The heads of the two queues stay in registers: one LDS
  // read per pick (the new head) instead of re-reading both heads and both picked weights — this loop is the longest
  // serial stretch of the encode kernel
  if (lane == 0) {
    constexpr uint32_t INF = 0xFFFFFFFFu;
    uint32_t li = 0, ii = m, ni = m;
    uint32_t wl = m ? hs->w[0] : INF, wi = INF;  // weight at the head of the leaf queue / the internal-node queue
    for (uint32_t step = 0; step + 1 < m; step++) {
      uint32_t pick0, pick1, v0, v1;
      {
        const bool tl = li < m && (ii >= ni || wl <= wi);
        if (tl) { pick0 = li++; v0 = wl; wl = li < m ? hs->w[li] : INF; }
        else { pick0 = ii++; v0 = wi; wi = ii < ni ? hs->w[ii] : INF; }
      }
      {
        const bool tl = li < m && (ii >= ni || wl <= wi);
        if (tl) { pick1 = li++; v1 = wl; wl = li < m ? hs->w[li] : INF; }
        else { pick1 = ii++; v1 = wi; wi = ii < ni ? hs->w[ii] : INF; }
      }
      hs->w[ni] = v0 + v1;
      hs->parent[pick0] = (uint16_t)ni; hs->parent[pick1] = (uint16_t)ni;
      if (ii == ni) wi = v0 + v1;  // the internal queue was empty: the new node is its head
      ni++;
    }
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

// canonical codes (bit-reversed for LSB-first packing); one wavefront.  cnt: >= 16 u32 of LDS scratch (codes per length).
// A symbol's code = next[len] + (#earlier symbols with the same length)  (RFC 1951 3.2.2); the running next code of every length
// is wave-uniform and stays in scalar registers: no LDS round trip between the blocks of 64 symbols.
__device__ void huff_codes_wave(const uint8_t* lens, uint32_t n, uint16_t* codes, uint32_t* cnt) {
  const uint32_t lane = lane_id();
  if (lane < 16) cnt[lane] = 0;
  wave_sync();
  for (uint32_t s = lane; s < n; s += 64) if (lens[s]) atomicAdd(&cnt[lens[s]], 1u);
  wave_sync();
  uint32_t run[16];
  uint32_t code = 0;
  run[0] = 0;
#pragma unroll
  for (uint32_t b = 1; b <= 15; b++) { code = (code + (b > 1 ? uni32(cnt[b - 1]) : 0u)) << 1; run[b] = code; }
  for (uint32_t b0 = 0; b0 < n; b0 += 64) {
    const uint32_t s = b0 + lane;
    const uint32_t l = s < n ? lens[s] : 0u;
    uint32_t mycode = 0;
#pragma unroll
    for (uint32_t b = 1; b <= 15; b++) {
      const uint64_t mask = __ballot(l == b);
      if (l == b) mycode = run[b] + (uint32_t)__builtin_popcountll(mask & lanemask_lt());
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

// Code-length RLE of one tree, closed form per run (identical tokens to the oracle's loop):
//   zero run R:      R/138 x (18,138), then rem>=11 -> (18,rem) | rem>=3 -> (17,rem) | rem literal zeros
//   non-zero run R:  the value, then (R-1)/6 x (16,6), then rem>=3 -> (16,rem) | rem literals
// One wavefront; tokens are appended at sm->nr (lane 0 keeps the count).
template <typename SM>
__device__ void rle_tree_wave(const uint8_t* l, uint32_t n, SM* sm, uint32_t tok_base, uint32_t* tok_end) {
  const uint32_t lane = lane_id();
  uint64_t starts[5];
  uint32_t val[5];
#pragma unroll
  for (int b = 0; b < 5; b++) {
    const uint32_t i = b * 64 + lane;
    const uint32_t v = i < n ? l[i] : 0xFFu;
    const uint32_t pv = (i > 0 && i < n) ? l[i - 1] : 0x1FFu;
    val[b] = v;
    starts[b] = __ballot(i < n && (i == 0 || v != pv));
  }
  uint32_t base = tok_base;
#pragma unroll
  for (int b = 0; b < 5; b++) {
    const uint32_t i = b * 64 + lane;
    const bool is_start = (starts[b] >> lane) & 1ull;
    // next run start after i (or n)
    uint32_t nxt = n;
    {
      bool found = false;
#pragma unroll
      for (int c = 0; c < 5; c++) {
        if (c < b) continue;
        uint64_t m = starts[c];
        if (c == b) m &= (lane == 63) ? 0ull : (~0ull << (lane + 1));
        if (!found && m) { nxt = c * 64 + (uint32_t)__builtin_ctzll(m); found = true; }
      }
    }
    const uint32_t v = val[b];
    uint32_t R = is_start ? nxt - i : 0u;
    uint32_t cnt = 0, full = 0, rem = 0;
    if (is_start) {
      if (v == 0) { full = R / 138; rem = R % 138; cnt = full + (rem >= 3 ? 1u : rem); }
      else { const uint32_t r1 = R - 1; full = r1 / 6; rem = r1 % 6; cnt = 1 + full + (rem >= 3 ? 1u : rem); }
    }
    // exclusive scan of cnt over the wave
    const uint32_t inc = wave_incl_scan(cnt);
    const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    uint32_t at = base + inc - cnt;
    if (is_start) {
      if (v == 0) {
        for (uint32_t k = 0; k < full; k++) { sm->rle_sym[at] = 18; sm->rle_eb[at] = 7; sm->rle_ev[at] = 127; at++; }
        if (rem >= 11) { sm->rle_sym[at] = 18; sm->rle_eb[at] = 7; sm->rle_ev[at] = (uint8_t)(rem - 11); at++; }
        else if (rem >= 3) { sm->rle_sym[at] = 17; sm->rle_eb[at] = 3; sm->rle_ev[at] = (uint8_t)(rem - 3); at++; }
        else for (uint32_t k = 0; k < rem; k++) { sm->rle_sym[at] = 0; sm->rle_eb[at] = 0; sm->rle_ev[at] = 0; at++; }
      } else {
        sm->rle_sym[at] = (uint8_t)v; sm->rle_eb[at] = 0; sm->rle_ev[at] = 0; at++;
        for (uint32_t k = 0; k < full; k++) { sm->rle_sym[at] = 16; sm->rle_eb[at] = 2; sm->rle_ev[at] = 3; at++; }
        if (rem >= 3) { sm->rle_sym[at] = 16; sm->rle_eb[at] = 2; sm->rle_ev[at] = (uint8_t)(rem - 3); at++; }
        else for (uint32_t k = 0; k < rem; k++) { sm->rle_sym[at] = (uint8_t)v; sm->rle_eb[at] = 0; sm->rle_ev[at] = 0; at++; }
      }
    }
    base += tot;
  }
  *tok_end = base;
  wave_sync();
}

#ifdef HMSE_DFL_STAMPS
// diagnostic build only: per-phase shader-clock totals of thread 0, summed over jobs and workgroups
__device__ unsigned long long g_dfl_stamps[6][24];   // [size group + 3 * dictionary jobs][phase]
__device__ unsigned long long g_enc_stamps[8];   // encode kernel: clocks of thread 0 per phase (0 load, 1 trees, 2 rle+cl+decide, 3 codes, 4 emit, 5 copy-out), [7] records
#define STAMP(i) do { if (threadIdx.x == 0) { const unsigned long long now__ = clock64(); stamp_acc[i] += now__ - stamp_last; stamp_last = now__; } } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

// Per-chunk record in the workspace (written by the match kernel, read by the encode kernel):
//   [lf u32[288]][df u32[32]][ntok u32 ...] | tok u32[L] | chunks with a dictionary only: DELTA stream slot (L + 5, + 16 slack)
// Tokens in parse order, one dword each: low half < 256 = literal byte; low half >= 256 = match of length
// (low-253) at distance (high half).
// ONE record per chunk (round 3; two per chunk with a dictionary and a separate slot before: 6.25 -> 4.25 x the stored bytes on
// the headline corpus).  The dictionary job and — if the delta is no quick accept — the plain job of the same chunk use the same
// histogram and token area one after the other (the launches are ordered: dictionary jobs, their encodes, then the plain jobs).
// The FULL stream is written OVER the token list once the encode kernel has turned all of it into its LDS bit image (the
// stream never exceeds L + 5 <= 4 * L bytes); the DELTA stream has its own slot behind the tokens, where it survives the plain job.
__host__ __device__ __forceinline__ uint32_t rec_ntok_off() { return 1280u; }
__host__ __device__ __forceinline__ uint32_t rec_tok_off() { return 1296u; }
__host__ __device__ __forceinline__ uint32_t rec_body(uint32_t L) { return 1296u + 4u * ((L + 3u) & ~3u); }   // histograms + tokens; 16-byte aligned
// stream slots start 16-byte aligned: the encode kernel copies its bit image out in whole 16-byte stores, which stay inside the
// token area (round_up_16(L + 5) <= 4 * round_up_4(L)) or the DELTA slot (16 bytes of slack)
__host__ __device__ __forceinline__ uint32_t rec_slot_off(uint32_t L, uint32_t variant) { return variant ? rec_body(L) : rec_tok_off(); }
__host__ __device__ __forceinline__ uint32_t rec_size(uint32_t L, bool has_dict) { return (rec_body(L) + (has_dict ? L + 5u + 16u : 0u) + 255u) & ~255u; }

struct Args {
  const uint8_t* data; uint64_t n;
  const uint64_t* cuts; const uint64_t* chunk_ids; const int64_t* base; uint64_t n_sel;
  uint32_t depth;
  uint32_t base_is_chunk;    // base[] holds chunk indices (into cuts) instead of indices into the selection
  const uint64_t* rec_off;   // [n_sel] byte offset of chunk k's record
  uint8_t* recs;
  uint64_t rec_cap; uint32_t* status;
  uint32_t* len_full; uint32_t* len_delta;
  uint8_t* scratch; size_t scratch_stride;   // !LDSM only
  uint8_t* scratch2; size_t scratch2_stride; // match lengths/distances of the classes that keep them out of LDS
  const uint32_t* jobs; const uint32_t* n_jobs; uint32_t* counter;  // this class's job list
  unsigned long long* prof_ctr; uint32_t prof_slot;                  // diagnostics: tokens written (match) / read (encode) by this launch
};


// per-workgroup global scratch of the big class
struct Scratch {
  uint16_t S[65536]; uint8_t K[65536 + 16];
  uint16_t jumpA[32768 + 8];
};

// DICT: the instantiation that runs the dictionary jobs of its class (chunk + base chunk; yields the DELTA and the FULL
// record).  Plain jobs run the DICT = false instantiation, which carries none of the snapshot code or its registers.
template <int NT, int TCAP, int LCAP_, bool LDSM, bool MDG = false, bool MLG = false, bool NOK = false, bool DICT = false>
__global__ __launch_bounds__(NT, (LDSM ? (NT == 1024 && TCAP <= TCAP_SG2 ? 8 : 4) : 2)) void l1_deflate_kernel(Args a) {
  using LY = Layout<NT, TCAP, LCAP_, LDSM, MDG, MLG, NOK>;
  constexpr int LCAP = LY::LCAP;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* const W = smem + LY::W_OFF;
  uint32_t* const cur = (uint32_t*)(smem + LY::CUR_OFF);
  uint32_t* const mark = (uint32_t*)(smem + LY::MARK_OFF);
  Small<NT, !LDSM>& sm = *(Small<NT, !LDSM>*)(smem + LY::SMALL_OFF);
  Scratch* const sc = LDSM ? nullptr : (Scratch*)(a.scratch + (size_t)blockIdx.x * a.scratch_stride);
  uint16_t* const mdist_g = (uint16_t*)(a.scratch2 + (size_t)blockIdx.x * a.scratch2_stride);  // used by S2/SG/B only
  uint16_t* const S = LDSM ? (uint16_t*)(smem + LY::A_OFF) : sc->S;
  uint16_t* const jump = LDSM ? (uint16_t*)(smem + LY::A_OFF) : sc->jumpA;
  uint8_t* const K = LDSM ? (uint8_t*)(smem + LY::K_OFF) : sc->K;
  const uint32_t n_jobs = *a.n_jobs;
#ifdef HMSE_DFL_STAMPS
  unsigned long long stamp_acc[24] = {0}, stamp_last = clock64();
#endif

  // the whole chain at once (thread 0; the first job of a workgroup, and behind a job that was refused)
  auto pf_fetch_sync = [&]() {
    const uint32_t ji2 = atomicAdd(a.counter, 1u);
    sm.nx.ji = ji2;
    if (ji2 >= n_jobs) return;
    const uint32_t job2 = a.jobs[ji2];
    const uint64_t k2 = job2 >> 1;
    const uint64_t c2 = a.chunk_ids ? a.chunk_ids[k2] : k2;
    const uint64_t cs = a.cuts[c2];
    sm.nx.job = job2; sm.nx.c = c2; sm.nx.cstart = cs; sm.nx.L = (uint32_t)(a.cuts[c2 + 1] - cs); sm.nx.rec_at = a.rec_off[k2];
    if constexpr (DICT) {
      const int64_t bsel = a.base[k2];
      const uint64_t bc = (a.chunk_ids && !a.base_is_chunk) ? a.chunk_ids[bsel] : (uint64_t)bsel;
      uint64_t ds = a.cuts[bc], dl = a.cuts[bc + 1] - ds;
      if (dl > WMAX) { ds += dl - WMAX; dl = WMAX; }
      sm.nx.dstart = ds; sm.nx.Dl = (uint32_t)dl;
    }
  };
  if (threadIdx.x == 0) pf_fetch_sync();
  for (;;) {
    // (the thread index re-read behind a compiler barrier per job: nothing derived from it is hoisted out of this persistent
    // loop and kept in registers across the matcher — see the encode kernel)
    uint32_t t = threadIdx.x;
    asm volatile("" : "+v"(t));
    const uint32_t lane = t & 63u, wave = t >> 6;
    // The job's metadata (list entry -> chunk id -> cuts -> dictionary chunk -> its cuts; record offset) is a chain of five to seven
    // dependent global loads: 2-8 % of a job when it is walked at the job's start.  Thread 0 walks the NEXT job's chain one link per
    // phase of THIS job (PF_STAGE below: a link is issued at one phase boundary and consumed at the next, long after it has
    // arrived) and leaves the result in sm.nx; a job starts by reading sm.nx.
    // (Measured, round 4: a barrier here that waits for LDS traffic only — the finished job's global stores need not have drained —
    // changes nothing: 39.4 against 39.5 ms of plain kernels at 2 GB; the second workgroup of the CU fills the wait.)
    __syncthreads();
    const uint32_t ji = uni32(sm.nx.ji);   // wave-uniform, and said so: lengths and loop bounds live in SGPRs
    if (ji >= n_jobs) break;
    STAMP(10);
    const uint32_t job = uni32(sm.nx.job);
    const uint64_t k = job >> 1;
    constexpr uint32_t variant = DICT ? 1u : 0u;  // (== job & 1: the lists are split by variant)
    const uint64_t c = uni64(sm.nx.c);
    const uint64_t cstart = uni64(sm.nx.cstart);
    const uint32_t L = uni32(sm.nx.L);
    const uint32_t Dl = variant ? uni32(sm.nx.Dl) : 0u;
    const uint64_t dstart = variant ? uni64(sm.nx.dstart) : 0ull;
    (void)c;
    uint32_t* len_out = variant ? a.len_delta : a.len_full;
    const uint32_t T = Dl + L;
    // links of the next job's chain in flight (thread 0): issued at one PF_STAGE, consumed at the next; nothing is in flight
    // across the matcher's state machine (its registers are the kernel's peak)
    uint32_t pf_u = 0; uint64_t pf_a = 0, pf_b = 0, pf_c = 0;
    auto pf_stage = [&](int stg) {
      const bool valid = stg <= 1 || sm.nx.ji < n_jobs;
      switch (stg) {
        case 0: pf_u = atomicAdd(a.counter, 1u); break;
        case 1: sm.nx.ji = pf_u; pf_u = pf_u < n_jobs ? a.jobs[pf_u] : 0u; break;
        case 2: {
          sm.nx.job = pf_u;
          if (valid) { const uint64_t k2 = pf_u >> 1; pf_a = a.chunk_ids ? a.chunk_ids[k2] : k2; pf_b = a.rec_off[k2]; if (DICT) pf_c = (uint64_t)a.base[k2]; }
          break;
        }
        case 3: sm.nx.c = pf_a; sm.nx.rec_at = pf_b; if (DICT) sm.nx.dstart = pf_c; break;   // (dstart: the base index, until stage 6)
        case 4:
          if (valid) {
            const uint64_t c2 = sm.nx.c;
            pf_a = a.cuts[c2]; pf_b = a.cuts[c2 + 1];
            if (DICT) { const uint64_t bsel = sm.nx.dstart; pf_c = (a.chunk_ids && !a.base_is_chunk) ? a.chunk_ids[bsel] : bsel; }
          }
          break;
        case 5:
          sm.nx.cstart = pf_a; sm.nx.L = (uint32_t)(pf_b - pf_a);
          if (DICT && valid) { const uint64_t bc = pf_c; pf_a = a.cuts[bc]; pf_b = a.cuts[bc + 1]; }
          break;
        default:
          if (DICT && valid) { uint64_t ds = pf_a, dl = pf_b - pf_a; if (dl > WMAX) { ds += dl - WMAX; dl = WMAX; } sm.nx.dstart = ds; sm.nx.Dl = (uint32_t)dl; }
          break;
      }
    };
#define PF_STAGE(n) do { if (t == 0) pf_stage(n); } while (0)
    // job record: histograms and the token list go to the encode kernel through it.  Classes S2/SG/B keep the match
    // distances (S2) or lengths and distances (SG, B) in a per-workgroup global array while matching.
    const uint64_t rec_at = uni64(sm.nx.rec_at);
    uint8_t* const rec = a.recs + rec_at;
    uint16_t* const mdist = (LDSM && !MDG) ? (uint16_t*)(smem + LY::MD_OFF) : mdist_g;
    uint8_t* const mlen = (LDSM && !MLG) ? (uint8_t*)(smem + LY::ML_OFF) : (uint8_t*)(mdist_g + LCAP);
    // A dictionary job yields the chunk's DELTA record only (round 3, rule 7 of the oracle): the FULL record of a chunk with a
    // base is produced by a plain job in a second pass, and only when the delta turns out larger than a fifth of the chunk.
    if (L > (uint32_t)LCAP || T > (uint32_t)TCAP || rec_at + (uint64_t)rec_size(L, variant != 0) > a.rec_cap) {
      if (t == 0) { len_out[k] = 0xFFFFFFFFu; if (L <= 32768u) atomicOr(a.status, 2u); }  // record area too small
      __syncthreads();                 // everybody has read sm.nx
      if (t == 0) pf_fetch_sync();     // (no phases to hide the chain behind)
      continue;
    }
    const uint8_t* csrc = a.data + cstart;
    const uint8_t* dsrc = a.data + dstart;

    // ---- phase 0: window into LDS, clear tables -----------------------------------------------
    for (uint32_t i = t * 16; i < Dl; i += NT * 16) {
      if (i + 16 <= Dl) { const uint4 v = load_u4_unaligned(dsrc + i); __builtin_memcpy(W + i, &v, 16); }
      else for (uint32_t b = i; b < Dl; b++) W[b] = dsrc[b];
    }
    {
      // the chunk behind the dictionary: whole 16-byte LDS slots are written ALIGNED (the global side takes the misalignment: Dl is any
      // byte count, and a 16-byte LDS store off its alignment is replayed lane by lane), the edges byte by byte
      const uint32_t a0 = (Dl + 15u) & ~15u, a1 = T & ~15u;      // aligned slots [a0, a1) lie inside [Dl, T)
      if (a0 < a1) {
        for (uint32_t o = a0 + t * 16; o < a1; o += NT * 16) { const uint4 v = load_u4_unaligned(csrc + (o - Dl)); *(uint4*)(W + o) = v; }
        if (t < a0 - Dl) W[Dl + t] = csrc[t];
        if (t < T - a1) W[a1 + t] = csrc[a1 - Dl + t];
      } else if (t < L) W[Dl + t] = csrc[t];                     // (fewer than 31 bytes)
    }
    if (t < 64) W[T + t] = 0;
    for (uint32_t i = t; i < NBK / 2; i += NT) cur[i] = 0;
    for (uint32_t i = t; i < 288; i += NT) sm.lf[i] = 0;
    if (t < 32) sm.df[t] = 0;
    if (t == 0) sm.qhead = 0;
    if (t < 4) sm.ocnt[t] = 0;
    // (16 bytes per store: both arrays start on a 16-byte boundary and hold LCAP = a multiple of 16 bytes)
    for (uint32_t i = t * 16; i < L; i += NT * 16) *(uint4*)(mlen + i) = make_uint4(0, 0, 0, 0);
    __syncthreads();

    STAMP(0);
    PF_STAGE(0);
    // ---- phase 1-2: bucket histogram + exclusive scan -> bucket starts --------------------------
    const uint32_t nh = T >= 4 ? T - 3 : 0;
    // (four consecutive positions per lane from two ALIGNED dwords: a 4-byte LDS read at an arbitrary byte address is replayed lane by lane,
    // ~64 clocks per wave-instruction — §6.2 — and this loop, like the scatter below, read every position that way until round 4)
    for (uint32_t q4 = t * 4u; q4 < nh; q4 += NT * 4u) {
      const uint32_t d0 = *(const uint32_t*)(W + q4), d1 = *(const uint32_t*)(W + q4 + 4u);
      cur_inc(cur, hash4(d0));
      if (q4 + 1u < nh) cur_inc(cur, hash4(alignb(d1, d0, 1u)));
      if (q4 + 2u < nh) cur_inc(cur, hash4(alignb(d1, d0, 2u)));
      if (q4 + 3u < nh) cur_inc(cur, hash4(alignb(d1, d0, 3u)));
    }
    __syncthreads();
    {
      constexpr int PERW = NBK / 2 / NT > 0 ? NBK / 2 / NT : 1;  // packed words per thread
      uint32_t wv[PERW], sum = 0;
#pragma unroll
      for (int i = 0; i < PERW; i++) {
        const uint32_t idx = t * PERW + i;
        wv[i] = idx < NBK / 2 ? cur[idx] : 0u;
        sum += (wv[i] & 0xFFFFu) + (wv[i] >> 16);
      }
      uint32_t total;
      uint32_t ex = block_exclusive_scan<NT>(sum, sm.red, &total);
#pragma unroll
      for (int i = 0; i < PERW; i++) {
        const uint32_t idx = t * PERW + i;
        const uint32_t lo16 = ex; ex += wv[i] & 0xFFFFu;
        const uint32_t hi16 = ex; ex += wv[i] >> 16;
        if (idx < NBK / 2) cur[idx] = lo16 | (hi16 << 16);
      }
    }
    __syncthreads();
    STAMP(1);
    PF_STAGE(1);
    // ---- phase 3-4: scatter, then rank inside the bucket -> ascending positions -------------------
    {
      // Counting sort with ORDERED buckets, in place.  Positions are scattered NT at a time in ascending
      // order, so bucket contents are already ordered between steps; inside one step the same-hash
      // positions land in the contiguous slot range [before, after) of their bucket in arbitrary order
      // and are ranked there (a handful of entries: the quadratic rank is over the step's duplicates
      // only, not over the whole bucket).  Two barriers per step: the next step's read of the cursors (which needs this
      // step's increments, complete at the first barrier) sits in front of the second one, so that barrier also separates
      // it from the next step's increments; the in-place rewrite of this step's slots [before, after) then runs beside the
      // next step's scatter into slots >= after.
      // (Measured, round 3: two or four positions per thread and step — half / a quarter of the barriers — are SLOWER, +4.5 % / +14 %
      // on the match kernels: the quadratic rank runs over the step's duplicates, and a step of 2048 positions has four times the pairs.)
      // (filter byte of a position: low nibble of its byte 4 | four more bits of its hash product, see the matcher's `rejects`)
      uint32_t q = t, h = 0, before = 0, hx = 0;
      bool act = q < nh;
      if (act) { hx = sort_key(W, q); h = hx >> (32 - HB); before = cur_get(cur, h); }
      __syncthreads();
      for (uint32_t q0 = 0; q0 < nh; q0 += NT) {
        if (act) S[cur_inc(cur, h)] = (uint16_t)q;
        __syncthreads();
        uint32_t r = 0;
        if (act) {
          const uint32_t after = cur_get(cur, h);
          for (uint32_t jj = before; jj < after; jj++) r += S[jj] < q;   // (measured: four slots per round trip is slower, +3 % — the usual trip count is one)
        }
        const uint32_t qn = q0 + NT + t;
        const bool actn = qn < nh;
        uint32_t hn = 0, beforen = 0, hxn = 0;
        if (actn) { hxn = sort_key(W, qn); hn = hxn >> (32 - HB); beforen = cur_get(cur, hn); }
        __syncthreads();
        if (act) { S[before + r] = (uint16_t)q; if constexpr (!NOK) K[before + r] = (uint8_t)((hx & 0x0Fu) | ((hx >> 12) & 0xF0u)); }
        q = qn; act = actn; h = hn; before = beforen; hx = hxn;
      }
      __syncthreads();  // cursor h now = end of bucket h
    }
    STAMP(2);
    PF_STAGE(2);
    // ---- phase 4b (dictionary jobs): diagonal anchors ------------------------------------------------
    // A chunk with an LSH base is a near-duplicate of it: most positions have a 258-byte match in the dictionary on the
    // same diagonal as their neighbours, and comparing those bytes again at every position is most of a dictionary job's
    // matcher time.  So one position in 64 (an ANCHOR) looks for a long dictionary match cooperatively — the wavefront's
    // lanes take the bucket's nearest dictionary members, then compare 512 bytes along the best one's diagonal at once —
    // and records (distance d, run length).  That is a true statement about the window, "W[x] == W[x - d] for x in
    // [anchor, anchor + run), and not at anchor + run", from which the matcher below knows the exact common prefix of any
    // position p in that run with its candidate p - d WITHOUT reading a byte.  The match results stay those of the
    // definition (the oracle's serial walk): a hint only replaces one candidate's byte compare by its known outcome.
    uint32_t* const anch = mark;   // u32[ceil(L / 64)]: d | run << 16, 0 = none (DICT only)
    if constexpr (DICT) {
      const uint32_t n_anch = (L + 63u) >> 6;
      for (uint32_t a0 = wave; a0 < n_anch; a0 += NT / 64) {
        const uint32_t pa = Dl + (a0 << 6);
        uint32_t word = 0;
        if (pa + 4 <= T) {
          const uint32_t h = hash4(ld32a(W, pa));
          const uint32_t lo = h ? cur_get(cur, h - 1) : 0u, hi = cur_get(cur, h);
          uint32_t l = lo, r = hi;   // bucket members ascend: dictionary positions first; [lo, l) after the search
          while (l < r) { const uint32_t m = (l + r) >> 1; if ((uint32_t)S[m] < Dl) l = m + 1; else r = m; }
          const uint32_t first = l > lo + 64u ? l - 64u : lo;   // the (up to) 64 nearest dictionary members
          const uint32_t idx = first + lane;
          uint32_t ml = 0, q = 0;
          if (idx < l) {
            q = S[idx];
            if (TCAP <= (int)WMAX || pa - q <= WMAX) {
              uint64_t qa[4], pb[4];
              ldNa<4>(W, q, qa); ldNa<4>(W, pa, pb);
              ml = 32;
#pragma unroll
              for (int u = 3; u >= 0; u--) { const uint64_t x = qa[u] ^ pb[u]; if (x) ml = 8u * u + ((uint32_t)__builtin_ctzll(x) >> 3); }
            }
          }
          uint32_t key = (ml << 8) | lane;   // longest, then nearest (highest lane)
          key = wave_max(key);
          if ((key >> 8) >= 32u) {
            const uint32_t d = pa - (uint32_t)__builtin_amdgcn_readlane((int)q, (int)(key & 63u));
            // 512 bytes along the diagonal, 8 per lane
            const uint32_t off = 8u * lane;
            uint32_t n = 8;
            if (pa + off < T) {
              uint64_t xa[1], xb[1];
              ldNa<1>(W, pa + off - d, xa); ldNa<1>(W, pa + off, xb);
              const uint64_t x = xa[0] ^ xb[0];
              if (x) n = (uint32_t)__builtin_ctzll(x) >> 3;
            } else n = 0;
            const uint64_t mm = __ballot(n < 8u);
            uint32_t run = 512;
            if (mm) { const uint32_t fl = (uint32_t)__builtin_ctzll(mm); run = 8u * fl + (uint32_t)__builtin_amdgcn_readlane((int)n, (int)fl); }
            if (run > T - pa) run = T - pa;
            word = d | (run << 16);
          }
        }
        anch[a0] = word;   // every lane, same value
      }
      __syncthreads();
    }
    // ---- phase 5: longest match for every chunk position ----------------------------------------
    // Bucket sizes and match lengths are heavily skewed (a few hot 4-grams hold most candidates), so a
    // lane does NOT own a fixed set of positions: every lane is a small state machine that pulls the
    // next sorted rank from an LDS counter when it has finished one, and advances by one unit of work
    // per iteration (fetch / probe one candidate / extend a long compare by 8 bytes).  Per candidate:
    // one u16 read of the bucket array and one 4-byte probe at offset best-3 (a longer match must agree
    // there); the first 16 bytes of the p side are compared from registers.  Candidate order per
    // position is unchanged (nearest first), so results equal the oracle's serial walk.
    //
    // Dictionary jobs (rule 2c of the oracle): a position with a diagonal hint (>= 16 bytes of an anchor's run left) TAKES it — no
    // walk.  Those are ~85 % of a near-duplicate's positions; a position-parallel pre-pass writes their results with coalesced stores,
    // and the state machine below only ever sees the others: every wavefront scans the sorted ranks 64 at a time, keeps the
    // ranks that need a walk (chunk positions without a hint) as a 64-bit window mask, and hands them to its idle
    // lanes in order — no lane ever pulls a dictionary rank or a hinted position.
    // diagonal hint of chunk position p: best (known length, candidate) over this block's anchor and the previous one's
    auto hint_of = [&](uint32_t p, uint32_t maxlen, uint32_t& bq) -> uint32_t {
      const uint32_t a0 = (p - Dl) >> 6;
      uint32_t bm = 0; bq = 0;
#pragma unroll
      for (uint32_t back = 0; back < 2; back++) {
        if (a0 < back) continue;
        const uint32_t w = anch[a0 - back];
        const uint32_t d = w & 0xFFFFu, run = w >> 16, pa = Dl + ((a0 - back) << 6);
        if (w == 0 || p < pa || d > p || p - d >= Dl) continue;      // (p < pa cannot happen: this block's anchor IS p's block start)
        uint32_t m = 0;
        if (run >= 512u) m = maxlen;                                   // no mismatch within 512 bytes: >= 385 from here
        else if (pa + run > p) m = (pa + run - p) < maxlen ? (pa + run - p) : maxlen;
        if (m > bm) { bm = m; bq = p - d; }
      }
      return bm;
    };
    if constexpr (DICT) {
      for (uint32_t x = t; x < L; x += NT) {
        const uint32_t p = Dl + x;
        if (p + 4 > T) continue;
        const uint32_t maxlen = (T - p) < MAXM ? (T - p) : MAXM;
        uint32_t bq;
        const uint32_t bm = hint_of(p, maxlen, bq);
        if (bm >= 16u) { mlen[x] = (uint8_t)(bm - 3); mdist[x] = (uint16_t)(p - bq); }
      }
      // (no barrier needed: the state machine never touches a hinted position's slots, and the parse starts behind a barrier)
    }
    // ---- phase 4c (LDS classes): the matcher's hand-out order ---------------------------------------------------------
    // A walk lasts about as many trips as its position has candidates (in-bucket index, capped at the depth), candidates per position
    // are heavily skewed, and a job is over when its LAST walk is: handed out in rank order, the deep members of the last buckets start
    // when the queue is nearly empty and every wavefront ends in a long, thinly occupied tail (a third of all lane-slots in the
    // scheduling model of tools/lz_sched.py, 19 % measured).  So the ranks are handed out longest-first: four classes by candidate
    // count (>= 24, 12..23, 4..11, 1..3), each a dense list of ranks; first-of-bucket ranks (no candidate: a third of all ranks) are
    // not handed out at all.  The lists live in the workgroup's global scratch (no LDS left in class S): two u16[32768] buffers, two
    // classes each, one growing up and one growing down.  Model: wave-trips per job 45.3 -> 34.1, pulls 20.3 -> 13.2 on wiki-synth;
    // results unchanged (the order of the walks does not enter them).
    // Dictionary jobs: only chunk positions WITHOUT a diagonal hint are listed (rule 2c: the others were written by the pre-pass above);
    // a job has ~2.6 of them per lane, so its matcher phase is as long as its longest walks make it — started first here.
#if defined(HMSE_NO_ORD)
    constexpr bool ORD = false;   // (A/B build: round 3's hand-out in rank order)
#elif defined(HMSE_NO_DICT_ORD)
    constexpr bool ORD = !DICT && LDSM;
#else
    constexpr bool ORD = LDSM;
#endif
    uint32_t* const ord = (uint32_t*)(a.scratch2 + (size_t)blockIdx.x * a.scratch2_stride + 98304u);
    if constexpr (ORD) {
      // Two passes over this wavefront's ranks (64 per step, a step every NT ranks; ITER steps at most): the first classifies — all
      // its LDS reads are independent and issue back to back — and counts per class in scalars; ONE LDS atomic per wavefront and class
      // reserves the list slots, and a barrier later the class totals place the four lists one behind the other; the second pass (no
      // memory reads: what it stores sits packed in registers, 10 bits per rank) writes the list.
      // (A first version with one atomic round trip per step cost 5.5 % of a job: seven dependent LDS chains per wavefront.)
      // A list entry carries what the matcher's pull would otherwise compute again for every position, ~15 vector instructions issued
      // for the whole wavefront in nearly every trip: rank | candidate count << 15 | the four hash-product bits of the filter byte << 21.
      constexpr int ITER = (TCAP + NT - 1) / NT;
      const uint32_t wv = uni32(wave);   // (scalar control: the cross-lane reads below never run under a narrowed EXEC)
      uint32_t pk[(ITER + 2) / 3];
#pragma unroll
      for (int w = 0; w < (ITER + 2) / 3; w++) pk[w] = 0;
      uint32_t n0 = 0, n1 = 0, n2 = 0, n3 = 0;
      // (branch-free and staged — all S reads, then all window reads, then all cursor reads: three LDS round trips per wavefront
      // instead of three per step; as nested ifs the compiler serialised the steps, 27 dependent round trips in class S)
      {
        const uint32_t last = nh ? nh - 1u : 0u;
        constexpr int G = DICT ? 2 : ITER > 12 ? 4 : 8;   // steps staged together (registers: the larger classes run under a 64-VGPR cap)
#pragma unroll
        for (int g = 0; g < ITER; g += G) {
          uint32_t pv[G], hv[G], lv[G];
#pragma unroll
          for (int u = 0; u < G; u++) if (g + u < ITER) { const uint32_t r = (wv << 6) + (uint32_t)(g + u) * NT + lane; pv[u] = S[r < last ? r : last]; }
#pragma unroll
          for (int u = 0; u < G; u++) if (g + u < ITER) hv[u] = ld32a(W, pv[u]) * 0x9E3779B1u;   // (the hash product: bucket = its top HB bits)
#pragma unroll
          for (int u = 0; u < G; u++) if (g + u < ITER) lv[u] = cur_get(cur, ((hv[u] >> (32 - HB)) - 1u) & (uint32_t)(NBK - 1));   // (= end of the bucket before; unused for bucket 0)
#pragma unroll
          for (int u = 0; u < G; u++) if (g + u < ITER) {
            const int it = g + u;
            const uint32_t r = (wv << 6) + (uint32_t)it * NT + lane;
            uint32_t km = r - ((hv[u] >> (32 - HB)) ? lv[u] : 0u);
            km = km < a.depth ? km : a.depth;
            km = km < 63u ? km : 63u;   // (6 bits in the entry; a deeper configuration recomputes the count at the pull)
            km = r < nh ? km : 0u;
            if constexpr (DICT) {
              const uint32_t pp = pv[u];
              uint32_t bq;
              const uint32_t bm = pp >= Dl ? hint_of(pp, (T - pp) < MAXM ? (T - pp) : MAXM, bq) : 16u;
              km = bm >= 16u ? 0u : km;   // a dictionary position (a candidate only) or a hinted chunk position: no walk
            }
            pk[it / 3] |= (km | ((hv[u] >> 10) & 0x3C0u)) << (10 * (it % 3));   // km (6 bits) | bits 16..19 of the hash product << 6
            n0 += (uint32_t)__builtin_popcountll(__ballot(km >= 24u)); n1 += (uint32_t)__builtin_popcountll(__ballot(km >= 12u && km < 24u));
            n2 += (uint32_t)__builtin_popcountll(__ballot(km >= 4u && km < 12u)); n3 += (uint32_t)__builtin_popcountll(__ballot(km >= 1u && km < 4u));
          }
        }
      }
      uint32_t bv = 0;
      if (lane < 4u) bv = atomicAdd(&sm.ocnt[lane], lane == 0u ? n0 : lane == 1u ? n1 : lane == 2u ? n2 : n3);
      uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)bv, 0), b1 = (uint32_t)__builtin_amdgcn_readlane((int)bv, 1),
               b2 = (uint32_t)__builtin_amdgcn_readlane((int)bv, 2), b3 = (uint32_t)__builtin_amdgcn_readlane((int)bv, 3);
      __syncthreads();   // every wavefront's counts are in: class totals -> where each class's list starts
      {
        const uint32_t c0 = uni32(sm.ocnt[0]), c1 = uni32(sm.ocnt[1]), c2 = uni32(sm.ocnt[2]);
        b1 += c0; b2 += c0 + c1; b3 += c0 + c1 + c2;
      }
#pragma unroll
      for (int it = 0; it < ITER; it++) {
        const uint32_t r0 = (wv << 6) + (uint32_t)it * NT;
        if (r0 < nh) {
          const uint32_t kv = (pk[it / 3] >> (10 * (it % 3))) & 0x3FFu, km = kv & 63u;
          const uint64_t m0 = __ballot(km >= 24u), m1 = __ballot(km >= 12u && km < 24u), m2 = __ballot(km >= 4u && km < 12u), m3 = __ballot(km >= 1u && km < 4u);
          if (km != 0u) {
            const uint64_t mm = km >= 24u ? m0 : km >= 12u ? m1 : km >= 4u ? m2 : m3;
            const uint32_t pos = (km >= 24u ? b0 : km >= 12u ? b1 : km >= 4u ? b2 : b3) + mbcnt64(mm);
            ord[pos] = (r0 + lane) | (kv << 15);
          }
          b0 += (uint32_t)__builtin_popcountll(m0); b1 += (uint32_t)__builtin_popcountll(m1);
          b2 += (uint32_t)__builtin_popcountll(m2); b3 += (uint32_t)__builtin_popcountll(m3);
        }
      }
      __syncthreads();
      STAMP(8);
    }
    STAMP(6);
    PF_STAGE(3);
    {
      enum { FETCH = 0, PROBE = 1, EXTEND = 2, DONE = 3 };
#ifndef HMSE_FETCH_BATCH
#define HMSE_FETCH_BATCH 16
#endif
#ifndef HMSE_FETCH_BATCH_DICT
#define HMSE_FETCH_BATCH_DICT 8
#endif
      constexpr uint32_t FETCH_BATCH = DICT ? HMSE_FETCH_BATCH_DICT : HMSE_FETCH_BATCH;
      uint32_t st = FETCH, i = 0, p = 0, kk = 0, kmax = 0, best = 0, bd = 0, probe = 0, maxlen = 0, ml = 0, q = 0, qn = 0, kn = 0;
      uint32_t pw0 = 0, pw1 = 0;
#ifndef HMSE_NO_PROBE16
      uint32_t pw2 = 0, pw3 = 0;   // bytes 8..15 of the position: a candidate that agrees on 8 bytes is compared on 8 more in the same trip
#endif
      uint32_t pkey = 0;   // classes with the filter array: this position's own filter byte
      // Can candidate with filter byte kb be skipped without a window read?  Filter array classes: kb = (byte 4 & 15) | 4 further
      // bits of the hash product << 4 — a candidate whose high nibble differs holds a DIFFERENT 4-gram in the same bucket
      // (a quarter of all candidates on text) and can never match; one whose low nibble differs has a different byte 4 and cannot
      // beat a best >= 4.  Class SG2/SG3 (no array): kb = the candidate's byte 4, read from the window.
      auto rejects = [&](uint32_t kb) -> bool {
        if constexpr (NOK) return best >= 4 && kb != (pw1 & 0xFFu);
        else { const uint32_t x = kb ^ pkey; return (x & 0xF0u) != 0 || (best >= 4 && (x & 0x0Fu) != 0); }
      };
      // dictionary jobs: this wavefront's window of pending work ranks (wave-uniform)
      uint64_t wq_mask = 0; uint32_t wq_base = 0; bool wq_done = false;
      const uint32_t nh_s = uni32(nh);
      // plain jobs: ranks per class -> ends of the four list segments in hand-out order
      const uint32_t n_ord = ORD ? uni32(sm.ocnt[0]) + uni32(sm.ocnt[1]) + uni32(sm.ocnt[2]) + uni32(sm.ocnt[3]) : nh_s;
      uint8_t* const wtab = sm.wtab + (wave << 6);
      // state of a freshly pulled position (rank ii holds chunk position pp)
      auto begin_walk = [&](uint32_t ii, uint32_t pp) {
        i = ii; p = pp;
        qn = i ? S[i - 1] : 0u;  // first candidate (used iff kmax != 0)
        kn = NOK ? (uint32_t)W[qn + 4] : (i ? (uint32_t)K[i - 1] : 0u);  // its byte 4 (class SG2 has no filter array)
#ifndef HMSE_NO_PROBE16
        { uint64_t pp__[2]; ldNa<2>(W, p, pp__); pw0 = (uint32_t)pp__[0]; pw1 = (uint32_t)(pp__[0] >> 32); pw2 = (uint32_t)pp__[1]; pw3 = (uint32_t)(pp__[1] >> 32); }
#else
        ld64a(W, p, pw0, pw1);
#endif
        const uint32_t hxp = pw0 * 0x9E3779B1u;
        const uint32_t h = hxp >> (32 - HB);
        pkey = (pw1 & 0x0Fu) | ((hxp >> 12) & 0xF0u);
        const uint32_t lo = h ? cur_get(cur, h - 1) : 0u;
        maxlen = (T - p) < MAXM ? (T - p) : MAXM;
        kmax = i - lo;
        if (kmax > a.depth) kmax = a.depth;
        best = MINM - 1; bd = 0; probe = pw0; kk = 1;
#ifdef HMSE_DFL_STAMPS
        if (DICT && t == 0) stamp_acc[7]++;  // lane 0's walked positions
#endif
        if (kmax != 0) st = PROBE;  // (else: first of its bucket, no match — the lane pulls again)
      };
      // the same from a hand-out list entry (rank | candidate count << 15 | filter bits << 21; a listed rank has a candidate, so i >= 1)
      auto begin_walk_listed = [&](uint32_t e) {
        i = e & 0x7FFFu; kmax = (e >> 15) & 63u;
        p = S[i];
        qn = S[i - 1];
        kn = NOK ? (uint32_t)W[qn + 4] : (uint32_t)K[i - 1];
#ifndef HMSE_NO_PROBE16
        { uint64_t pp__[2]; ldNa<2>(W, p, pp__); pw0 = (uint32_t)pp__[0]; pw1 = (uint32_t)(pp__[0] >> 32); pw2 = (uint32_t)pp__[1]; pw3 = (uint32_t)(pp__[1] >> 32); }
#else
        ld64a(W, p, pw0, pw1);
#endif
        pkey = (pw1 & 0x0Fu) | ((e >> 17) & 0xF0u);
        if (kmax == 63u) {   // (the entry's count saturates at 63: a deeper configuration recomputes it — no lane comes here at depth <= 62)
          const uint32_t h = hash4(pw0), lo = h ? cur_get(cur, h - 1) : 0u;
          kmax = i - lo < a.depth ? i - lo : a.depth;
        }
        maxlen = (T - p) < MAXM ? (T - p) : MAXM;
        best = MINM - 1; bd = 0; probe = pw0; kk = 1;
#ifdef HMSE_DFL_STAMPS
        if (DICT && t == 0) stamp_acc[7]++;
#endif
        st = PROBE;
      };
      // every wavefront leaves this loop: by running out of work, or — never observed on a correct build — by using up a trip
      // budget no legal walk can reach (status bit 4: the job's record is then garbage and the call reports it)
      const uint32_t trip_budget = (nh + 64u) * (a.depth + 16u);
      uint32_t trips = 0;
      for (;;) {
#ifdef HMSE_DFL_STAMPS
        if (t == 0) stamp_acc[13]++;  // trips of wavefront 0 through the state machine
#endif
        if (++trips > uni32(trip_budget)) {
#ifdef HMSE_DIAG
          const uint64_t sf = __ballot(st == FETCH), sp = __ballot(st == PROBE), se = __ballot(st == EXTEND);
          if (lane == 0) printf("[dfl] trip budget: job %u k %llu L %u Dl %u nh %u wave %u qhead %u wq_base %u wq_mask %llx wq_done %d FETCH %llx PROBE %llx EXTEND %llx p %u kk %u kmax %u best %u ml %u maxlen %u q %u\n",
                                ji, (unsigned long long)k, L, Dl, nh, wave, sm.qhead, wq_base, (unsigned long long)wq_mask, (int)wq_done, (unsigned long long)sf, (unsigned long long)sp, (unsigned long long)se, p, kk, kmax, best, ml, maxlen, q);
#endif
          if (lane == 0) atomicOr(a.status, 16u);
          break;
        }
        uint64_t need = uni64(__ballot(st == FETCH));
        if constexpr (DICT && !ORD) {
          // serve the idle lanes from the wavefront's window; scan the next 64 sorted ranks when it is empty.  Everything that
          // steers this loop (need, wq_mask, wq_base, wq_done, nh_s) is wave-uniform AND scalar (uni32 / uni64)
          for (int it = 0; need != 0 && it < 6; it++) {
            if (wq_mask == 0) {
              if (wq_done) break;
              uint32_t b = 0;
              if (lane == 0) b = atomicAdd(&sm.qhead, 64u);
              b = uni32(b);
              if (b >= nh_s) { wq_done = true; break; }
              wq_base = b;
              const uint32_t ii = b + lane;
              bool work = false;
              if (ii < nh_s) {
                const uint32_t pp = S[ii];
                if (pp >= Dl) {
                  const uint32_t mx = (T - pp) < MAXM ? (T - pp) : MAXM;
                  uint32_t bq;
                  const uint32_t bm = hint_of(pp, mx, bq);
                  work = bm < 16u;
                }
              }
              wq_mask = uni64(__ballot(work));
              if (wq_mask == 0) continue;
            }
            const uint32_t navail = (uint32_t)__builtin_popcountll(wq_mask), nneed = (uint32_t)__builtin_popcountll(need);
            const uint32_t take_n = navail < nneed ? navail : nneed;
            const bool has = (wq_mask >> lane) & 1ull;
            const uint32_t rm = mbcnt64(wq_mask);
            if (has && rm < take_n) wtab[rm] = (uint8_t)lane;
            wave_sync();
            const bool isneed = (need >> lane) & 1ull;
            const uint32_t rn = mbcnt64(need);
            uint32_t got = 0xFFFFFFFFu;
            if (isneed && rn < take_n) got = wq_base + (uint32_t)wtab[rn];
            wave_sync();
            wq_mask &= ~uni64(__ballot(has && rm < take_n));
            need &= ~uni64(__ballot(isneed && rn < take_n));
            if (got != 0xFFFFFFFFu) begin_walk(got, (uint32_t)S[got]);
          }
          if (wq_done && wq_mask == 0 && st == FETCH) st = DONE;
        } else if (need && ((uint32_t)__builtin_popcountll(need) >= FETCH_BATCH || uni64(__ballot(st == PROBE || st == EXTEND)) == 0)) {
          // wave-aggregated pull of the next sorted ranks — once FETCH_BATCH lanes are idle (or nothing else can run): the pull's
          // ~40 instructions are issued for the whole wavefront whatever the number of lanes they serve (measured, 10 GB: plain
          // kernels 180.8 ms at 1, 179.3 at 8, 178.2 at 16, 180.1 at 24, 184.5 at 32)
          const uint32_t leader = (uint32_t)__builtin_ctzll(need);
          uint32_t base = 0;
          if (lane == leader) base = atomicAdd(&sm.qhead, (uint32_t)__builtin_popcountll(need));
          base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)leader);   // leader is wave-uniform: v_readlane, no LDS crossbar trip
          if (st == FETCH) {
            const uint32_t f = base + mbcnt64(need);
            if (f >= n_ord) st = DONE;
            else if constexpr (ORD) begin_walk_listed(ord[f]);
            else begin_walk(f, (uint32_t)S[f]);
          }
        }
        if (uni64(__ballot(st != DONE)) == 0) break;
#ifdef HMSE_DFL_STAMPS
        const uint64_t occP__ = __ballot(st == PROBE), occE__ = __ballot(st == EXTEND);
        if (t == 0) {   // lane occupancy of the matcher's blocks in this trip (wavefront 0)
          const uint32_t nP = (uint32_t)__builtin_popcountll(occP__), nE = (uint32_t)__builtin_popcountll(occE__);
          const uint32_t nF = (uint32_t)__builtin_popcountll(need);
          stamp_acc[16] += nF != 0; stamp_acc[17] += nF; stamp_acc[18] += nP != 0; stamp_acc[19] += nP; stamp_acc[20] += nE != 0; stamp_acc[21] += nE;
        }
#endif
        bool fin = false;  // candidate kk finished with length ml
#ifdef HMSE_EXT_MIN
        // (experiment) the EXTEND block's ~120 instructions are issued for the whole wavefront however few lanes extend: run it only when
        // HMSE_EXT_MIN lanes wait for it, or nothing else can run
        const uint64_t em__ = uni64(__ballot(st == EXTEND));
        const bool run_ext = (uint32_t)__builtin_popcountll(em__) >= (uint32_t)HMSE_EXT_MIN || uni64(__ballot(st == PROBE)) == 0;
#else
        constexpr bool run_ext = true;
#endif
        if (st == PROBE) {
          q = qn;
          uint32_t kb = kn;
          if (kk < kmax) { qn = S[i - kk - 1]; kn = NOK ? (uint32_t)W[qn + 4] : (uint32_t)K[i - kk - 1]; }  // prefetch the next candidate
          // a candidate the byte-4 filter rejects (37 % of them on text) is consumed on the spot and the next one takes
          // its place in this trip: the filter needs nothing but the two prefetched values
          // (classes with the filter array only: where byte 4 is a dependent window read the second test costs more than it saves)
          if (!NOK && rejects(kb) && kk < kmax && !(TCAP > (int)WMAX && p - q > WMAX) ) {
            kk++;
            q = qn; kb = kn;
            if (kk < kmax) { qn = S[i - kk - 1]; kn = NOK ? (uint32_t)W[qn + 4] : (uint32_t)K[i - kk - 1]; }
          }
          fin = true; ml = 0;
          if (TCAP > (int)WMAX && p - q > WMAX) kk = kmax;             // farther ones are farther still
          else if (rejects(kb)) { }                                    // another 4-gram, or byte 4 differs: at most 4 <= best
          else {
            // random-address window reads only for candidates that can still win
            const uint32_t cprobe = ld32a(W, q + best - 3);
            uint32_t c0, c1;
            ld64a(W, q, c0, c1);
            uint32_t x;
            if (cprobe != probe) { }                                   // cannot beat the current best
            else if ((x = c0 ^ pw0) != 0) ml = (uint32_t)__builtin_ctz(x) >> 3;
            else if ((x = c1 ^ pw1) != 0) ml = 4 + ((uint32_t)__builtin_ctz(x) >> 3);
#ifndef HMSE_NO_PROBE16
            else if (maxlen > 8) {
              // 8 bytes agree: 8 more at once (95 % of such candidates end there on text: tools/lz_study "extend trips at 32 / 16 / 8 B"),
              // so the EXTEND block — ~55 vector instructions issued for the whole wavefront, for ~4 lanes — is entered by the rare
              // candidate that agrees on 16 bytes, and a match of 8..15 bytes costs no trip of its own
              uint32_t c2, c3;
              ld64a(W, q + 8, c2, c3);
              if ((x = c2 ^ pw2) != 0) ml = 8 + ((uint32_t)__builtin_ctz(x) >> 3);
              else if ((x = c3 ^ pw3) != 0) ml = 12 + ((uint32_t)__builtin_ctz(x) >> 3);
              else { ml = 16; if (maxlen > 16) { fin = false; st = EXTEND; } }
            } else ml = 8;
#else
            else { ml = 8; if (maxlen > 8) { fin = false; st = EXTEND; } }
#endif
          }
        } else if (st == EXTEND && run_ext) {
          // 32 bytes per trip, all reads of each side issued together (one LDS round trip): the dictionary jobs
          // (near-duplicate chunks) spend most of their trips here, a full-length match is 258 bytes
          // (64 bytes where the class runs one workgroup per CU and so may use 128 VGPRs: SG3 and B, the classes of the
          // largest dictionary jobs)
          if constexpr (!LDSM || TCAP > TCAP_SG2) {
            uint64_t da[8], db[8];
            ldNa<8>(W, q + ml, da); ldNa<8>(W, p + ml, db);
            uint32_t adv = 64;
            bool hit = false;
#pragma unroll
            for (int u = 7; u >= 0; u--) { const uint64_t x = da[u] ^ db[u]; if (x) { adv = 8 * u + ((uint32_t)__builtin_ctzll(x) >> 3); hit = true; } }
            ml += adv;
            if (hit || ml >= maxlen) fin = true;
          } else {
            uint64_t qa[4], pa[4];
            ldNa<4>(W, q + ml, qa); ldNa<4>(W, p + ml, pa);
            const uint64_t x = qa[0] ^ pa[0], y = qa[1] ^ pa[1], z = qa[2] ^ pa[2], w = qa[3] ^ pa[3];
            if (x) { ml += (uint32_t)__builtin_ctzll(x) >> 3; fin = true; }
            else if (y) { ml += 8 + ((uint32_t)__builtin_ctzll(y) >> 3); fin = true; }
            else if (z) { ml += 16 + ((uint32_t)__builtin_ctzll(z) >> 3); fin = true; }
            else if (w) { ml += 24 + ((uint32_t)__builtin_ctzll(w) >> 3); fin = true; }
            else { ml += 32; if (ml >= maxlen) fin = true; }
          }
        }
        if (fin) {
          if (ml > maxlen) ml = maxlen;
          bool pos_done = false;
          if (ml > best) {
            best = ml; bd = p - q;
            if (ml == maxlen) pos_done = true; else probe = ld32a(W, p + best - 3);
          }
          if (++kk > kmax) pos_done = true;
          st = PROBE;
          if (pos_done) {
            if (best >= MINM) { mlen[p - Dl] = (uint8_t)(best - 3); mdist[p - Dl] = (uint16_t)bd; }
            st = FETCH;
          }
        }
      }
    }
    __syncthreads();
    for (uint32_t i = t; i < (L >> 5) + 2; i += NT) mark[i] = 0;   // (held the diagonal anchors of a dictionary job until here)
    STAMP(3);
    PF_STAGE(4);
    const uint8_t* mlen_c = mlen;
    // Match lengths that live in HBM (the classes that keep them out of LDS) are staged into a free LDS region first: the
    // parse reads each of them three times.
    {
      uint8_t* stage = nullptr;
      if constexpr (!LDSM) stage = smem + LY::A_OFF;                                   // unused by this kernel in class B
      else if constexpr (MLG) {
        const uint32_t joff = (2u * (L + 1u) + 15u) & ~15u;                            // behind the next-pointers
        if (joff + L + 16u <= (uint32_t)LY::A_SZ) stage = smem + LY::A_OFF + joff;
      }
      if (stage) {
        __syncthreads();
        for (uint32_t i = t * 16; i < L; i += NT * 16) { const uint4 v = *(const uint4*)(mlen_c + i); *(uint4*)(stage + i) = v; }
        mlen_c = stage;
        __syncthreads();
      }
    }
    const uint16_t* const mdist_c = mdist;
    uint8_t* const rec_c = rec;
    // ---- phase 6: parse by pointer doubling -------------------------------------------------------
    auto take = [&](uint32_t x) -> bool {
      const uint32_t ml = mlen_c[x];
      return ml != 0 && !(x + 1 < L && mlen_c[x + 1] > ml);
    };
    for (uint32_t x = t; x <= L; x += NT) {
      uint32_t nx = L;
      if (x < L) { nx = x + (take(x) ? (uint32_t)mlen_c[x] + 3u : 1u); if (nx > L) nx = L; }
      jump[x] = (uint16_t)nx;
    }
    __syncthreads();
    STAMP(11);
    PF_STAGE(5);
    // The parse chain 0 -> next(0) -> ... is serial, but LZ parses re-synchronise: chains started at
    // different positions usually merge after a few tokens.  So every wavefront walks its own segment
    // speculatively from the segment start (64 positions per step: the window's `next` values sit in one
    // VGPR and the walk inside the window is a chain of v_readlane, no memory latency), then wave 0
    // stitches the segments in order: from the true entry of a segment it walks only until it meets a
    // speculatively marked position, keeps the speculative marks from there on and drops the ones before.
    {
      constexpr uint32_t NW = NT / 64;
      const uint32_t SEG = (((L + NW - 1) / NW) + 63u) & ~63u;
      {
        const uint32_t sw = wave * SEG;
        const uint32_t send = (sw + SEG) < L ? (sw + SEG) : L;
        uint32_t curp = sw;
        // Pointer doubling inside a 64-position window instead of a serial chain: after round r lane j knows the first 2^r
        // positions of the path that starts at j (a 64-bit mask) and where that path stands (its 2^r-th successor, or
        // the position at which it left the window).  <= 6 rounds of three lane gathers; the walk itself is then two
        // v_readlane at the entry.  The doubling of a window does not depend on the walk's entry into it, so PWX windows can be
        // doubled TOGETHER (independent gathers: one LDS-crossbar round trip serves PWX windows) and the entries resolved afterwards,
        // in order.  Measured, round 4: three windows together are 1.8 % SLOWER on the plain kernels (38.8 -> 39.5 ms at 2 GB) —
        // the walks' latency is already hidden behind the CU's other workgroup, the extra registers and selects are not.
#ifndef HMSE_DFL_PW
#define HMSE_DFL_PW 1
#endif
        constexpr int PWX = HMSE_DFL_PW;
        const uint32_t sw_s = uni32(sw), send_s = uni32(send);   // (wave-uniform, and said so: the window loop is a scalar loop)
        for (uint32_t wb0 = sw_s; wb0 < send_s; wb0 += 64u * PWX) {
          uint32_t hop[PWX], rlo[PWX], rhi[PWX], wendv[PWX];
#pragma unroll
          for (int u = 0; u < PWX; u++) {
            const uint32_t wb = wb0 + 64u * u;
            const uint32_t x = wb + lane;
            hop[u] = (wb < send_s && x < L) ? (uint32_t)jump[x] : L;
            wendv[u] = (wb + 64) < send_s ? (wb + 64) : send_s;
            rlo[u] = lane < 32 ? 1u << lane : 0u; rhi[u] = lane >= 32 ? 1u << (lane - 32) : 0u;
          }
#pragma nounroll
          for (int r = 0; r < 6; r++) {
            bool inside[PWX], any = false;
#pragma unroll
            for (int u = 0; u < PWX; u++) { inside[u] = (wb0 + 64u * u) < send_s && hop[u] < wendv[u]; any = any || inside[u]; }
            if (__ballot(any) == 0) break;
            uint32_t h2[PWX], l2[PWX], g2[PWX];
#pragma unroll
            for (int u = 0; u < PWX; u++) {
              const int src = (int)(inside[u] ? hop[u] - (wb0 + 64u * u) : lane);
              h2[u] = (uint32_t)__shfl((int)hop[u], src, 64);
              l2[u] = (uint32_t)__shfl((int)rlo[u], src, 64); g2[u] = (uint32_t)__shfl((int)rhi[u], src, 64);
            }
#pragma unroll
            for (int u = 0; u < PWX; u++) {
              rlo[u] |= inside[u] ? l2[u] : 0u;
              rhi[u] |= inside[u] ? g2[u] : 0u;
              hop[u] = inside[u] ? h2[u] : hop[u];
            }
          }
#pragma unroll
          for (int u = 0; u < PWX; u++) {
            const uint32_t wb = wb0 + 64u * u;
            if (wb < send_s) {
              uint64_t m = 0;
              if (curp < wendv[u]) {
                const int el = (int)(curp - wb);
                m = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)rhi[u], el) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)rlo[u], el);
                curp = (uint32_t)__builtin_amdgcn_readlane((int)hop[u], el);
              }
              mark[wb >> 5] = (uint32_t)m; mark[(wb >> 5) + 1] = (uint32_t)(m >> 32);  // every lane, same words (no lane-dependent branch at a loop tail)
            }
          }
        }
        sm.pexit[wave] = curp;  // every lane, same value
      }
      __syncthreads();
      STAMP(12);
      // Stitch segment w from its true entry e: windows wholly before e hold no true position; from e's window on every
      // window is rewritten (a long match may jump over whole windows, whose speculative marks must go too) until the
      // walk meets a speculatively marked position, whose marks are kept from there on.  Returns the segment's true exit.
      // `chain_exit` = exit of the chain the segment's marks currently describe (returned when the walk meets it).
      auto stitch = [&](uint32_t w, uint32_t e, uint32_t chain_exit, bool intact, bool& conv_out) -> uint32_t {
        const uint32_t sw = w * SEG;
        const uint32_t send = (sw + SEG) < L ? (sw + SEG) : L;
        conv_out = true;
        if (intact && e == sw) return chain_exit;  // the speculation started at the true entry (marks untouched so far)
        for (uint32_t wb = sw; wb < send && (wb + 64 <= e || e >= send); wb += 64)
          { mark[wb >> 5] = 0; mark[(wb >> 5) + 1] = 0; }
        uint32_t curp = e;
        bool conv = false;
        for (uint32_t wb = e & ~63u; wb < send && !conv; wb += 64) {
          const uint32_t x = wb + lane;
          const uint32_t nv = x < L ? (uint32_t)jump[x] : L;
          const uint64_t spec = (uint64_t)mark[wb >> 5] | ((uint64_t)mark[(wb >> 5) + 1] << 32);
          const uint32_t wend = (wb + 64) < send ? (wb + 64) : send;
          uint64_t m = 0;
          while (curp < wend) {
            if ((spec >> (curp - wb)) & 1ull) { conv = true; break; }
            m |= 1ull << (curp - wb);
            curp = (uint32_t)__builtin_amdgcn_readlane((int)nv, (int)(curp - wb));
          }
          if (conv) m |= spec & ~((1ull << (curp - wb)) - 1ull);  // speculative marks from the meeting point on
          mark[wb >> 5] = (uint32_t)m; mark[(wb >> 5) + 1] = (uint32_t)(m >> 32);  // every lane, same words (no lane-dependent branch at a loop tail)
        }
        conv_out = conv;
        return conv ? chain_exit : curp;
      };
      // Second-level speculation: chains almost always meet inside a segment, and then the segment's exit is the
      // speculative one — so every wavefront stitches ITS segment at once, taking the previous segment's speculative exit
      // as entry.  Segment w's result is final iff all segments before it met their speculative chain; from the first
      // one that did not, wave 0 redoes the rest in order (the marks left by a stitch from a wrong entry are still
      // one successor-closed chain with exit pexit2[w], which is all the stitch relies on).
      {
        const uint32_t sw = wave * SEG;
        bool conv = true;
        uint32_t ex = sm.pexit[wave];
        if (wave > 0 && sw < L) ex = stitch(wave, sm.pexit[wave - 1], sm.pexit[wave], true, conv);
        wave_sync();
        sm.pexit2[wave] = ex; sm.pconv[wave] = (wave == 0 || sw >= L || conv) ? 1u : 0u;
      }
      __syncthreads();
      if (wave == 0) {
        uint32_t first_bad = NW;
        for (uint32_t w = 0; w < NW; w++) if (sm.pconv[w] == 0u) { first_bad = w; break; }
        if (first_bad < NW) {
          uint32_t tru = sm.pexit2[first_bad];  // exact: its entry was exact
          for (uint32_t w = first_bad + 1; w < NW; w++) {
            if (w * SEG >= L) break;
            bool cv;
            tru = stitch(w, tru, sm.pexit2[w], false, cv);  // pexit2[w]: exit of the chain the first pass left marked there
          }
        }
      }
    }
    __syncthreads();
    STAMP(4);
    PF_STAGE(6);
    // ---- phase 7: tokens in parse order + symbol histograms ------------------------------------------------
    // token index of a marked position = number of marked positions before it: workgroup prefix scan over the
    // popcounts of contiguous mark words, then every thread walks the set bits of its own words
    {
      uint32_t* const tok = (uint32_t*)(rec_c + rec_tok_off());
      // every thread owns ceil(L / NT) consecutive positions (not whole mark words: with one 32-position word per
      // thread only L/32 of the NT threads would have work, each with a serial run of ~9 tokens)
      const uint32_t ppt = (L + NT - 1) / NT;
      const uint32_t p0 = t * ppt < L ? t * ppt : L, p1 = (p0 + ppt) < L ? (p0 + ppt) : L;
      auto bits_at = [&](uint32_t pos, uint32_t nb) -> uint32_t {  // nb <= 32 mark bits from position pos on
        const uint32_t w = pos >> 5;
        const uint64_t v = (uint64_t)mark[w] | ((uint64_t)mark[w + 1] << 32);
        return (uint32_t)(v >> (pos & 31u)) & (nb >= 32u ? 0xFFFFFFFFu : ((1u << nb) - 1u));
      };
      uint32_t cntm = 0;
      for (uint32_t pos = p0; pos < p1; pos += 32) cntm += (uint32_t)__builtin_popcount(bits_at(pos, p1 - pos));
      uint32_t ntok;
      uint32_t idx = block_exclusive_scan<NT>(cntm, sm.red, &ntok);
      for (uint32_t pos = p0; pos < p1; pos += 32) {
        uint32_t m = bits_at(pos, p1 - pos);
        while (m) {
          const uint32_t x = pos + (uint32_t)__builtin_ctz(m);
          m &= m - 1;
          if (take(x)) {
            const uint32_t l3 = mlen_c[x], dd = mdist_c[x];
            uint32_t code, eb, ev;
            len_sym(l3 + 3u, code, eb, ev); atomicAdd(&sm.lf[code], 1u);
            dist_sym(dd, code, eb, ev); atomicAdd(&sm.df[code], 1u);
            tok[idx] = (256u + l3) | (dd << 16);
          } else {
            const uint32_t b = W[Dl + x];
            atomicAdd(&sm.lf[b], 1u);
            tok[idx] = b;
          }
          idx++;
        }
      }
      if (t == 0) { atomicAdd(&sm.lf[256], 1u); *(uint32_t*)(rec_c + rec_ntok_off()) = ntok; if (a.prof_ctr) atomicAdd(&a.prof_ctr[a.prof_slot], (unsigned long long)ntok); }
      __syncthreads();
      STAMP(5);
      uint32_t* const r_hist = (uint32_t*)rec_c;
      for (uint32_t i = t; i < 288; i += NT) r_hist[i] = sm.lf[i];
      if (t < 32) r_hist[288 + t] = sm.df[t];
    }
    STAMP(9);
#ifdef HMSE_DFL_STAMPS
    if (t == 0) { stamp_acc[14] += L; stamp_acc[15]++; }
#endif
  }
#undef PF_STAGE
#ifdef HMSE_DFL_STAMPS
  if (threadIdx.x == 0) for (int i = 0; i < 24; i++) atomicAdd(&g_dfl_stamps[(TCAP <= TCAP_S ? 0 : TCAP <= TCAP_SG ? 1 : 2) + (DICT ? 3 : 0)][i], stamp_acc[i]);
#endif
}


// ---- encode kernel: Huffman trees, block type, bit image -----------------------------------------------------------------
// One small workgroup per job record, many records per CU: the serial parts of one record (two-queue merge, stitching of the header)
// hide behind the others.
// Throughput = records in flight per CU / latency of one record, and a record's latency is mostly ONE lane's serial work (the two-queue
// merges).  Measured, round 4, on round 3's kernel (256 threads, the whole bit image in LDS: 17.5 KiB, 8 records per CU = the wave limit):
// padding its LDS so that 6 / 5 / 4 / 2 workgroups fit a CU takes it from 5.6 to 7.1 / 8.0 / 10.1 / 17.3 ms at 2 GB — time ~ 1 / records in
// flight (profiles/r4/r4_encode_records_in_flight_2GB.txt).  So this kernel holds SIXTEEN records per CU: two wavefronts per record (the
// wave limit again: 16 x 2 = 32) and 10 160 bytes of LDS (16 x 10 KiB = the CU's 160 KiB; 13 records for chunks above 12 KiB, whose batch
// offsets need 2 KiB more) — the bit image is a 4 KiB WINDOW that slides over the stream: the token batches (64 tokens, alternating between
// the two wavefronts) whose bits end inside the window are emitted, the finished 16-byte blocks are copied to the record's slot and the
// window moves on (a typical 8 KiB chunk's ~2.7 KiB stream needs one window).  Same stream, bit for bit: 5.6 -> 4.2 ms at 2 GB.
struct EncSmall {
  uint32_t lf[288], df[32], cf[20];   // lf: later the bit offset of every token batch (u32[193])
  uint8_t ll[288], dl[32], cl[20];
  uint16_t lc[288], dc[32], cc[20];
  uint8_t rle_sym[352], rle_eb[352], rle_ev[352];
  uint32_t nr, nlit, ndist, ncl, mode, hdr_bits, fixed_bits, extra_bits, data_bits, cl_bits;
  uint32_t job, spill;
};
template <int LMIN_, int LCAP_>
struct EncLayout {
  static constexpr int NT = 128, LMIN = LMIN_, LCAP = LCAP_;
  static constexpr int NB_MAX = (LCAP + 63) / 64;       // token batches of a record (a token per byte at most)
  static constexpr int NBLK = (NB_MAX + 1 + 63) / 64;   // 64-entry blocks of the batch-offset array
  static constexpr int SMALL_OFF = 0;
  static constexpr int SMALL_SZ = align16((int)sizeof(EncSmall));
  static constexpr int HS_OFF = SMALL_OFF + SMALL_SZ;
  static constexpr int HS_SZ = align16((int)(sizeof(HuffL) + sizeof(HuffD)));
  static constexpr int TT_OFF = HS_OFF;                 // u32 TT[512] and the image window reuse the Huffman scratch (dead once the codes exist)
  static constexpr int OUT_OFF = TT_OFF + 2048;
  static constexpr int WIN_BYTES = 4096;                // the window; behind it: the last put's overhang (<= 8 bytes) and the 16-byte block in progress
  static constexpr int OUT_SZ = HS_SZ - 2048;
  static constexpr bool BOFF_IN_LF = NB_MAX + 1 <= 288; // batch offsets: in the dead histogram when they fit, else in an array of their own
  static constexpr int BOFF_OFF = HS_OFF + HS_SZ;
  static constexpr int BOFF_SZ = BOFF_IN_LF ? 0 : align16(4 * (NB_MAX + 1));
  static constexpr int TOTAL = BOFF_OFF + BOFF_SZ;
  static_assert(OUT_SZ >= WIN_BYTES + 64, "window + overhang");
  static_assert(LCAP > 12288 || TOTAL <= 10240, "sixteen workgroups per CU for the records of chunks <= 12 KiB");
};

template <int LMIN, int LCAP>
__global__ __launch_bounds__(128, 8) void l1_encode_kernel(Args a, const bool force_spill) {
  using EL = EncLayout<LMIN, LCAP>;
  constexpr uint32_t NT = EL::NT;
  constexpr uint32_t WIN_BITS = EL::WIN_BYTES * 8;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  EncSmall& sm = *(EncSmall*)(smem + EL::SMALL_OFF);
  HuffL* const hsL = (HuffL*)(smem + EL::HS_OFF);
  HuffD* const hsD = (HuffD*)(smem + EL::HS_OFF + sizeof(HuffL));
  uint32_t* const TT = (uint32_t*)(smem + EL::TT_OFF);
  uint32_t* const out = (uint32_t*)(smem + EL::OUT_OFF);
  uint32_t* const boff = EL::BOFF_IN_LF ? sm.lf : (uint32_t*)(smem + EL::BOFF_OFF);
  const uint32_t n_jobs = *a.n_jobs;
#ifdef HMSE_DFL_STAMPS
  unsigned long long e_acc0 = 0, e_acc1 = 0, e_acc2 = 0, e_acc3 = 0, e_acc4 = 0, e_acc5 = 0, e_n = 0, e_last = clock64();
#define ESTAMP(v) do { if (threadIdx.x == 0) { const unsigned long long now__ = clock64(); v += now__ - e_last; e_last = now__; } } while (0)
#else
#define ESTAMP(v) do { } while (0)
#endif
  for (;;) {
  uint32_t t = threadIdx.x;
  // The thread index is re-read behind a compiler barrier in every round: otherwise everything derived from it (dozens of
  // `t < 288`-style masks, LDS addresses) is hoisted out of this persistent loop and held in registers across all phases and
  // spilled to scratch under the 64-register cap that eight wavefronts per SIMD need (tools/kernel_regs.py)
  asm volatile("" : "+v"(t));
  const uint32_t lane = t & 63u, wave = uni32(t >> 6);
  __syncthreads();
  if (t == 0) sm.job = atomicAdd(a.counter, 1u);
  __syncthreads();
  const uint32_t ji = uni32(sm.job);
  if (ji >= n_jobs) break;
  const uint32_t job = uni32(a.jobs[ji]);
  const uint64_t k = job >> 1;
  const uint32_t variant = job & 1u;
  const uint64_t c = a.chunk_ids ? uni64(a.chunk_ids[k]) : k;
  const uint64_t cstart = uni64(a.cuts[c]);
  const uint32_t L = uni32((uint32_t)(a.cuts[c + 1] - cstart));
  if (L <= (uint32_t)LMIN || L > (uint32_t)LCAP) continue;      // (lists are split by length: cannot happen)
  uint32_t* len_out = variant ? a.len_delta : a.len_full;
  if (uni32(len_out[k]) == 0xFFFFFFFFu) continue;           // the match kernel could not take this job
  uint8_t* const rec = a.recs + uni64(a.rec_off[k]);
  const uint32_t* const r_hist = (const uint32_t*)rec;
  const uint32_t ntok = uni32(*(const uint32_t*)(rec + rec_ntok_off()));
  const uint32_t* const tok = (const uint32_t*)(rec + rec_tok_off());
  if (a.prof_ctr && t == 0) atomicAdd(&a.prof_ctr[a.prof_slot], (unsigned long long)ntok);
  uint8_t* const slot = rec + rec_slot_off(L, variant);
  const uint8_t* const lit = a.data + cstart;
  for (uint32_t i = t; i < 288; i += NT) sm.lf[i] = r_hist[i];
  if (t < 32) sm.df[t] = r_hist[288 + t];
  if (t < 20) sm.cf[t] = 0;
  __syncthreads();
  ESTAMP(e_acc0);
  // ---- trees (wave 0: lit/len; wave 1: dist, then the fixed-code cost) ---------------------------------------------------
  if (wave == 0) huff_lengths_wave(sm.lf, 286, 15, sm.ll, hsL);
  else {
    huff_lengths_wave(sm.df, 30, 15, sm.dl, hsD);
    uint32_t fb = 0, xb = 0;
    for (uint32_t s = lane; s < 286; s += 64) { fb += sm.lf[s] * fixed_len(s); if (s >= 257) xb += sm.lf[s] * len_extra_bits(s); }
    if (lane < 30) { fb += sm.df[lane] * 5u; xb += sm.df[lane] * dist_extra_bits(lane); }
    fb = wave_sum(fb); xb = wave_sum(xb);
    if (lane == 0) { sm.fixed_bits = fb + xb + 3; sm.extra_bits = xb; }
  }
  __syncthreads();
  ESTAMP(e_acc1);
  // ---- code-length RLE + CL tree (wave 0), dynamic data bits (wave 1) -----------------------------------------------------
  if (wave == 0) {
    uint32_t nlit = 286, ndist = 30;
    {
      uint32_t hi = 0;
      for (uint32_t b = 0; b < 320; b += 64) { const uint32_t s = b + lane; const uint64_t m = __ballot(s < 286 && sm.ll[s] != 0); if (m) hi = b + 64 - (uint32_t)__builtin_clzll(m); }
      nlit = hi < 257 ? 257 : hi;
      const uint64_t md = __ballot(lane < 30 && sm.dl[lane] != 0);
      ndist = md ? 64 - (uint32_t)__builtin_clzll(md) : 1u;
    }
    uint32_t e1, e2;
    rle_tree_wave(sm.ll, nlit, &sm, 0, &e1);
    rle_tree_wave(sm.dl, ndist, &sm, e1, &e2);
    for (uint32_t i = lane; i < e2; i += 64) atomicAdd(&sm.cf[sm.rle_sym[i]], 1u);
    if (lane == 0) { sm.nlit = nlit; sm.ndist = ndist; sm.nr = e2; }
    wave_sync();
    huff_lengths_wave(sm.cf, 19, 7, sm.cl, hsD);
    const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    const uint64_t mo = __ballot(lane < 19 && sm.cl[order[lane < 19 ? lane : 0]] != 0);
    uint32_t ncl = mo ? 64 - (uint32_t)__builtin_clzll(mo) : 0u;
    if (ncl < 4) ncl = 4;
    uint32_t cb = 0;
    for (uint32_t i = lane; i < e2; i += 64) cb += sm.cl[sm.rle_sym[i]] + sm.rle_eb[i];
    cb = wave_sum(cb);
    if (lane == 0) { sm.ncl = ncl; sm.cl_bits = cb; }
  } else {
    uint32_t db = 0;
    for (uint32_t s = lane; s < 286; s += 64) db += sm.lf[s] * sm.ll[s];
    if (lane < 30) db += sm.df[lane] * sm.dl[lane];
    db = wave_sum(db);
    if (lane == 0) sm.data_bits = db;
  }
  __syncthreads();
  if (t == 0) {
    const uint32_t dhdr = 3 + 14 + 3 * sm.ncl + sm.cl_bits;
    const uint32_t dyn = dhdr + sm.extra_bits + sm.data_bits;
    const uint32_t stored = 8u * (5u + L);
    uint32_t mode;
    if (stored <= sm.fixed_bits && stored <= dyn) mode = 0;
    else if (sm.fixed_bits <= dyn) mode = 1;
    else mode = 2;
    sm.mode = mode;
    sm.hdr_bits = mode == 2 ? dhdr : 3u;
  }
  __syncthreads();
  const uint32_t mode = uni32(sm.mode);
  if (mode == 0) {
    if (t == 0) {
      slot[0] = 1; slot[1] = (uint8_t)L; slot[2] = (uint8_t)(L >> 8); slot[3] = (uint8_t)~L; slot[4] = (uint8_t)(~L >> 8);
      len_out[k] = 5 + L;
    }
    for (uint32_t x = t; x < L; x += NT) slot[5 + x] = lit[x];
    continue;
  }
  ESTAMP(e_acc2);
  // ---- code tables ------------------------------------------------------------------------------------------------------------
  if (mode == 1) {
    for (uint32_t s = t; s < 288; s += NT) sm.ll[s] = (uint8_t)fixed_len(s);
    if (t < 32) sm.dl[t] = 5;
    __syncthreads();
  }
  if (wave == 0) huff_codes_wave(sm.ll, mode == 1 ? 288 : 286, sm.lc, hsL->cnt);
  else {
    huff_codes_wave(sm.dl, mode == 1 ? 32 : 30, sm.dc, hsD->cnt);
    if (mode == 2) huff_codes_wave(sm.cl, 19, sm.cc, hsD->cnt);
  }
  __syncthreads();
  for (uint32_t i = t; i < (uint32_t)EL::OUT_SZ / 4; i += NT) out[i] = 0;
  // TT[low half of a token] = the bits of its first part (literal code | length code + extra bits: <= 20 bits) | bit count << 24;
  // df[distance code] = code | length << 16 (the histogram is dead).  The emit loops then do one table read per part.
  for (uint32_t ts = t; ts < 512u; ts += NT) {
    uint32_t e;
    if (ts < 256u) e = (uint32_t)sm.lc[ts] | ((uint32_t)sm.ll[ts] << 24);
    else {
      uint32_t code, eb, ev;
      len_sym(ts - 253u, code, eb, ev);
      const uint32_t l1 = sm.ll[code];
      e = (uint32_t)sm.lc[code] | (ev << l1) | ((l1 + eb) << 24);
    }
    TT[ts] = e;
  }
  if (t < 32) sm.df[t] = (uint32_t)sm.dc[t] | ((uint32_t)sm.dl[t] << 16);
  __syncthreads();
  ESTAMP(e_acc3);
  // ---- emit ---------------------------------------------------------------------------------------------------------------------
  auto tokbits = [&](uint32_t tk) -> uint32_t {
    const uint32_t ts = tk & 0xFFFFu;
    uint32_t b = TT[ts] >> 24;
    if (ts >= 256u) {
      uint32_t dcode, deb, dev;
      dist_sym(tk >> 16, dcode, deb, dev);
      b += (sm.df[dcode] >> 16) + deb;
    }
    return b;
  };
  const uint32_t nb = (ntok + 63u) >> 6;   // token batches; batch b belongs to wavefront b & 1
  // (a) bits per batch (the histogram is dead: boff = lf), four batches' loads in flight per wavefront
  for (uint32_t b = wave; b < nb; b += 8u) {
    uint32_t tk[4];
#pragma unroll
    for (uint32_t u = 0; u < 4u; u++) { const uint32_t i = ((b + 2u * u) << 6) + lane; tk[u] = i < ntok ? tok[i] : 0xFFFFFFFFu; }
#pragma unroll
    for (uint32_t u = 0; u < 4u; u++) {
      const uint32_t bits = wave_sum(tk[u] != 0xFFFFFFFFu ? tokbits(tk[u]) : 0u);
      if (lane == 0 && b + 2u * u < nb) boff[b + 2u * u] = bits;
    }
  }
  __syncthreads();
  // (b) exclusive scan -> the bit offset of every batch in the stream (boff[nb] = where the end-of-block code goes)
  if (wave == 0) {
    uint32_t carry = sm.hdr_bits;
#pragma unroll
    for (uint32_t kx = 0; kx < (uint32_t)EL::NBLK; kx++) {
      const uint32_t idx = kx * 64u + lane;
      const uint32_t v = idx < nb ? boff[idx] : 0u;
      const uint32_t inc = wave_incl_scan(v);
      if (idx < nb) boff[idx] = carry + inc - v;
      carry += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    }
    if (lane == 0) boff[nb] = carry;
    // A FULL record's stream is written OVER its token list (one record per chunk), window by window — while later tokens
    // are still to be read: safe as long as the stream bytes below a window edge do not outnumber the bytes of the tokens consumed so
    // far (4 per token), i.e. <= 32 bits per token on average from the start of the record.  That holds for every record we have seen
    // (the fixed code's dearest token is 31 bits; 64 literals buy 23 x 64 bits of slack) but is no theorem for a dynamic code, so it is
    // CHECKED here at every batch boundary a window could end at, and a record that fails it sends its finished windows to this
    // workgroup's 32 KiB of the match kernels' global scratch (idle now) and moves them into the slot when its last token is read.
    bool risky = false;
#pragma unroll
    for (uint32_t kx = 0; kx < (uint32_t)EL::NBLK; kx++) {
      const uint32_t idx = kx * 64u + lane;
      if (idx >= 1u && idx < nb) { const uint32_t bo = boff[idx]; risky = risky || (bo >= WIN_BITS - 3200u && (bo >> 3) > 256u * idx); }
    }
    const bool any = __ballot(risky) != 0ull;
    if (lane == 0) sm.spill = (variant == 0u && (any || force_spill)) ? 1u : 0u;
  }
  __syncthreads();
  // (c) header (wave 0; it lies inside the first window: <= 17 + 57 + 316 x 14 bits)
  if (wave == 0) {
    if (lane == 0) { put_bits(out, 0, 1, 1); put_bits(out, 1, mode, 2); }
    if (mode == 2) {
      const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
      if (lane == 0) { put_bits(out, 3, sm.nlit - 257, 5); put_bits(out, 8, sm.ndist - 1, 5); put_bits(out, 13, sm.ncl - 4, 4); }
      if (lane < sm.ncl) put_bits(out, 17 + 3 * lane, sm.cl[order[lane]], 3);
      const uint32_t nr = sm.nr, per = (nr + 63) / 64;
      const uint32_t i0 = lane * per, i1 = (i0 + per) < nr ? (i0 + per) : nr;
      uint32_t bits = 0;
      for (uint32_t i = i0; i < i1; i++) bits += sm.cl[sm.rle_sym[i]] + sm.rle_eb[i];
      const uint32_t inc = wave_incl_scan(bits);
      uint32_t off = 17 + 3 * sm.ncl + inc - bits;
      for (uint32_t i = i0; i < i1; i++) {
        const uint32_t s = sm.rle_sym[i], l = sm.cl[s], eb = sm.rle_eb[i];
        put_bits(out, off, (uint32_t)sm.cc[s] | ((uint32_t)sm.rle_ev[i] << l), l + eb);
        off += l + eb;
      }
    }
  }
  // (d) the window slides over the batches: [seg, e) = the batches that END inside the window that starts at base_bits
  const uint32_t end_bits = uni32(boff[nb]);
  const uint32_t eob_len = sm.ll[256];
  const bool spill = uni32(sm.spill) != 0u;
  uint8_t* const wdst = spill ? a.scratch2 + (size_t)blockIdx.x * 32768u : slot;   // where finished windows go (the last one always goes to the slot)
  uint32_t seg = 0, base_bits = 0;
  // this wavefront's batches are b = wave, wave + 2, ...; their tokens are loaded three batches ahead (the loads outlive the window moves: a
  // move overwrites consumed tokens only)
  auto ldtok = [&](uint32_t b) -> uint32_t { const uint32_t i = (b << 6) + lane; return i < ntok ? tok[i] : 0xFFFFFFFFu; };
  uint32_t nxt = wave;
  uint32_t pf0 = ldtok(nxt), pf1 = ldtok(nxt + 2u), pf2 = ldtok(nxt + 4u);
  for (;;) {
    uint32_t cnt = 0;
#pragma unroll
    for (uint32_t kx = 0; kx < (uint32_t)EL::NBLK; kx++) {
      const uint32_t idx = seg + 1u + kx * 64u + lane;
      cnt += (uint32_t)__builtin_popcountll(__ballot(idx <= nb && boff[idx <= (uint32_t)EL::NB_MAX ? idx : 0u] - base_bits <= WIN_BITS));
    }
    const uint32_t e = uni32(seg + cnt);   // (> seg unless seg == nb: a batch is <= 64 x 48 bits, the window starts < 128 bits before it)
    for (; nxt < e; nxt += 2u) {
      const uint32_t b = nxt;
      const uint32_t tk = pf0;
      pf0 = pf1; pf1 = pf2; pf2 = ldtok(b + 6u);
      uint32_t v1 = 0, nb1 = 0, v2 = 0, nb2 = 0;
      if (tk != 0xFFFFFFFFu) {
        const uint32_t ts = tk & 0xFFFFu;
        const uint32_t en = TT[ts];
        v1 = en & 0xFFFFFFu; nb1 = en >> 24;
        if (ts >= 256u) {
          uint32_t dcode, deb, dev;
          dist_sym(tk >> 16, dcode, deb, dev);
          const uint32_t d = sm.df[dcode], l2 = d >> 16;
          v2 = (d & 0xFFFFu) | (dev << l2); nb2 = l2 + deb;
        }
      }
      const uint32_t bb = nb1 + nb2;
      const uint32_t inc = wave_incl_scan(bb);
      const uint32_t off = uni32(boff[b]) - base_bits + inc - bb;
      if (bb) {   // one put of up to 48 bits: at most three dwords of the image
        const uint64_t v = (uint64_t)v1 | ((uint64_t)v2 << nb1);
        const uint32_t w = off >> 5, sh = off & 31u;
        atomicOr(&out[w], (uint32_t)v << sh);
        if (sh + bb > 32u) {
          const uint64_t rest = (v >> 1) >> (31u - sh);
          atomicOr(&out[w + 1], (uint32_t)rest);
          if (sh + bb > 64u) atomicOr(&out[w + 2], (uint32_t)(rest >> 32));
        }
      }
    }
    if (e >= nb) break;
    __syncthreads();
    // move the window: whole 16-byte blocks below the next batch's first bit go to the slot, the block in progress moves to the front
    const uint32_t nbase = uni32(boff[e]) & ~127u;
    const uint32_t nflush = (nbase - base_bits) >> 3;
    for (uint32_t i = t * 16; i < nflush; i += NT * 16) { const uint4 v = *(const uint4*)((const uint8_t*)out + i); *(uint4*)(wdst + (base_bits >> 3) + i) = v; }
    const uint4 tail = *(const uint4*)((const uint8_t*)out + nflush);
    __syncthreads();
    for (uint32_t i = t; i < (uint32_t)EL::OUT_SZ / 4; i += NT) out[i] = 0;
    __syncthreads();
    if (t == 0) *(uint4*)out = tail;
    __syncthreads();
    base_bits = nbase; seg = e;
  }
  __syncthreads();
  if (t == 0) put_bits(out, end_bits - base_bits, sm.lc[256], eob_len);
  __syncthreads();
  ESTAMP(e_acc4);
  const uint32_t nbytes = (end_bits + eob_len + 7) >> 3;
  if (spill) for (uint32_t i = t * 16; i < (base_bits >> 3); i += NT * 16) *(uint4*)(slot + i) = *(const uint4*)(wdst + i);   // (every token has been read)
  for (uint32_t i = t * 16; (base_bits >> 3) + i < nbytes; i += NT * 16) {  // the slot is 16-byte aligned; whole 16-byte stores stay inside it (rec_slot_off)
    const uint4 v = *(const uint4*)((const uint8_t*)out + i);
    *(uint4*)(slot + (base_bits >> 3) + i) = v;
  }
  if (t == 0) len_out[k] = nbytes;
  ESTAMP(e_acc5);
#ifdef HMSE_DFL_STAMPS
  e_n++;
#endif
  }
#ifdef HMSE_DFL_STAMPS
  if (threadIdx.x == 0) {
    atomicAdd(&g_enc_stamps[0], e_acc0); atomicAdd(&g_enc_stamps[1], e_acc1); atomicAdd(&g_enc_stamps[2], e_acc2); atomicAdd(&g_enc_stamps[3], e_acc3);
    atomicAdd(&g_enc_stamps[4], e_acc4); atomicAdd(&g_enc_stamps[5], e_acc5); atomicAdd(&g_enc_stamps[7], e_n);
  }
#endif
#undef ESTAMP
}

// Size classes of the match kernel (window T = dictionary + chunk; caps: HMSE_TCAP_* at the top).  Per-position arrays in LDS:
//   S  : T <= 10048  window, sorted ranks, byte-4 filter, match lengths and distances   (80 KiB, two per CU)
//   S2 : T <= 13952  as S, match distances in a per-workgroup global array              (80 KiB, two per CU)
//   SG : T <= 17408  as S2, match lengths there too                                     (80 KiB, two per CU)
//   SG2: T <= 22976  as SG, without the byte-4 filter array                             (80 KiB, two per CU)
//   SG3: T <= 32768  as SG2                                                             (112 KiB, one per CU)
//   B  : T <= 65536  window in LDS, everything else in a per-workgroup global scratch (dictionary jobs only; 1024 threads, one per CU)
// Two-per-CU classes run 1024 threads capped at 64 VGPRs: the CU's full 32 waves.  (Round 4: the caps were 9216 / 12288 / 16000 / 21504 with
// the encode kernel's fields in `Small`; a plain job costs the same per byte in the class below, so the plain kernels did not move, but 3 ms of
// dictionary jobs moved from the one-per-CU class SG3 to SG2.)
constexpr int N_LIST = 17;      // plain jobs per class: lists 0 (S), 4 (S2), 5 (SG), 8 (SG2), 2 (SG3), 3 (B); dictionary jobs per class:
                                // lists 9..14 (dict_list); encode-kernel lists by chunk length: 6 and 7 (FULL records), 15 and 16 (DELTA
                                // records).  The dictionary jobs run FIRST: the chunks whose delta is no quick accept (rule 7) then join
                                // the plain lists for their FULL record, so every kernel symbol is launched once per call
constexpr int N_CTR = 64;       // u32 counters: [c] = jobs in list c, [32 + c] = list c's cursor
__host__ __device__ constexpr uint32_t dict_list(uint32_t c) { return c == 0 ? 9u : c == 4 ? 10u : c == 5 ? 11u : c == 8 ? 12u : c == 2 ? 13u : 14u; }
__host__ __device__ __forceinline__ uint32_t size_class(uint64_t T) {
  return T <= (uint64_t)TCAP_S ? 0u : T <= (uint64_t)TCAP_S2 ? 4u : T <= (uint64_t)TCAP_SG ? 5u : T <= (uint64_t)TCAP_SG2 ? 8u : T <= (uint64_t)TCAP_SG3 ? 2u : 3u;
}
// rule 7 of the oracle: a delta of at most a fifth of the chunk (README.md:1328, 2175) is the record without a look at FULL
__host__ __device__ __forceinline__ bool delta_gate(uint32_t n2, uint64_t len, uint32_t pct) { return n2 != 0xFFFFFFFFu && n2 != 0u && !(pct && (uint64_t)n2 * 100 > (uint64_t)pct * len); }
__host__ __device__ __forceinline__ bool delta_quick(uint32_t n2, uint64_t len, uint32_t pct) { return delta_gate(n2, len, pct) && (uint64_t)n2 * 5 <= len; }
static_assert(2 * Layout<NT_S, TCAP_S, TCAP_S, true>::TOTAL <= 160 * 1024, "class S must fit twice per CU");
static_assert(2 * Layout<NT_S, TCAP_S2, TCAP_S2, true, true>::TOTAL <= 160 * 1024, "class S2 must fit twice per CU");
static_assert(2 * Layout<NT_S, TCAP_SG, TCAP_SG, true, true, true>::TOTAL <= 160 * 1024, "class SG must fit twice per CU");
static_assert(2 * Layout<NT_S, TCAP_SG2, TCAP_SG2, true, true, true, true>::TOTAL <= 160 * 1024, "class SG2 must fit twice per CU");

// workgroup-aggregated append of up to two jobs per thread to their lists: LDS counters, then ONE global atomic per list and
// workgroup (per wavefront before: 88 k same-address atomics per 10 GB shard on a handful of counters, 1.7 ms).  All threads call it.
constexpr int LIST_NT = 1024;
__device__ __forceinline__ void list_append2(bool on1, uint32_t c1, uint32_t job1, bool on2, uint32_t c2, uint32_t job2,
                                             uint32_t* lists, uint64_t list_stride, uint32_t* counts) {
  __shared__ uint32_t s_cnt[N_LIST], s_base[N_LIST];
  if (threadIdx.x < (uint32_t)N_LIST) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  uint32_t i1 = 0, i2 = 0;
  if (on1) i1 = atomicAdd(&s_cnt[c1], 1u);
  if (on2) i2 = atomicAdd(&s_cnt[c2], 1u);
  __syncthreads();
  if (threadIdx.x < (uint32_t)N_LIST) { const uint32_t n = s_cnt[threadIdx.x]; s_base[threadIdx.x] = n ? atomicAdd(&counts[threadIdx.x], n) : 0u; }
  __syncthreads();
  if (on1) lists[(size_t)c1 * list_stride + s_base[c1] + i1] = job1;
  if (on2) lists[(size_t)c2 * list_stride + s_base[c2] + i2] = job2;
}

// job = (k << 1) | variant, appended to its size class's list.  A chunk with a base gets a dictionary job (its DELTA record)
// and nothing else in this pass; a chunk without one a plain job (its FULL record).
__global__ __launch_bounds__(LIST_NT) void classify_kernel(const uint64_t* __restrict__ cuts, const uint64_t* __restrict__ chunk_ids,
                                                        const int64_t* __restrict__ base, uint64_t n_sel, uint32_t base_is_chunk,
                                                        uint32_t* __restrict__ lists, uint64_t list_stride, uint32_t* __restrict__ counts,
                                                        const uint64_t* __restrict__ n_dev) {
  if (n_dev) n_sel = *n_dev;   // captured chain: the count lives in HBM, the grid is sized for the worst case
  const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool in = k < n_sel;
  uint64_t L = 0; bool hasb = false; uint64_t Dl = 0;
  if (in) {
    const uint64_t c = chunk_ids ? chunk_ids[k] : k;
    L = cuts[c + 1] - cuts[c];
    if (base && base[k] >= 0) {
      const uint64_t bc = (chunk_ids && !base_is_chunk) ? chunk_ids[base[k]] : (uint64_t)base[k];
      Dl = cuts[bc + 1] - cuts[bc];
      if (Dl > WMAX) Dl = WMAX;
      hasb = true;
    }
  }
  const bool enc_ok = in && L <= 32768;
  const uint32_t job = (uint32_t)(k << 1) | (hasb ? 1u : 0u);
  list_append2(in, hasb ? dict_list(size_class(L + Dl)) : size_class(L), job,
               enc_ok, hasb ? (L <= 12288 ? 15u : 16u) : (L <= 12288 ? 6u : 7u), job, lists, list_stride, counts);
}

// rule 7: chunks with a base whose delta is no quick accept need their FULL record after all — they join the plain lists
__global__ __launch_bounds__(LIST_NT) void redo_kernel(const uint64_t* __restrict__ cuts, const uint64_t* __restrict__ chunk_ids,
                                                    const int64_t* __restrict__ base, uint64_t n_sel, const uint32_t* __restrict__ len_delta,
                                                    uint32_t pct, uint32_t* __restrict__ lists, uint64_t list_stride,
                                                    uint32_t* __restrict__ counts, const uint64_t* __restrict__ n_dev) {
  if (n_dev) n_sel = *n_dev;
  const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool redo = false; uint64_t L = 0;
  if (k < n_sel && base && base[k] >= 0) {
    const uint64_t c = chunk_ids ? chunk_ids[k] : k;
    L = cuts[c + 1] - cuts[c];
    redo = !delta_quick(len_delta[k], L, pct);
  }
  list_append2(redo, size_class(L), (uint32_t)(k << 1), redo && L <= 32768, L <= 12288 ? 6u : 7u, (uint32_t)(k << 1), lists, list_stride, counts);
}

// record sizes (one record per chunk; a chunk with a dictionary has the DELTA stream's slot behind its tokens)
__global__ __launch_bounds__(256) void rec_size_kernel(const uint64_t* __restrict__ cuts, const uint64_t* __restrict__ chunk_ids,
                                                        const int64_t* __restrict__ base, uint64_t n_sel,
                                                        uint64_t* __restrict__ sizes, const uint64_t* __restrict__ n_dev) {
  if (n_dev) n_sel = *n_dev;
  const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_sel) return;
  const uint64_t c = chunk_ids ? chunk_ids[k] : k;
  const uint64_t len = cuts[c + 1] - cuts[c];
  sizes[k] = len <= 32768 ? rec_size((uint32_t)len, base && base[k] >= 0) : 0u;
}

// kind decision (rule 7 of the oracle; README.md:1328, 2175) and final lengths
__global__ __launch_bounds__(256) void decide_kernel(const uint64_t* __restrict__ cuts, const uint64_t* __restrict__ chunk_ids,
                                                      const int64_t* __restrict__ base, uint64_t n_sel,
                                                      const uint32_t* __restrict__ len_full, const uint32_t* __restrict__ len_delta,
                                                      uint32_t pct, uint64_t* __restrict__ final_len, uint8_t* __restrict__ kind,
                                                      uint32_t* status, const uint64_t* __restrict__ n_dev) {
  if (n_dev) n_sel = *n_dev;
  const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_sel) return;
  uint32_t n = 0; uint8_t kd = HMSE_KIND_FULL;
  bool done = false;
  if (base && base[k] >= 0) {
    const uint32_t n2 = len_delta[k];
    const uint64_t c = chunk_ids ? chunk_ids[k] : k;
    const uint64_t len = cuts[c + 1] - cuts[c];
    if (delta_quick(n2, len, pct)) { n = n2; kd = HMSE_KIND_DELTA; done = true; }        // FULL was never computed
    else {
      const uint32_t n1 = len_full[k];
      if (n1 != 0xFFFFFFFFu && n1 != 0u && delta_gate(n2, len, pct) && (uint64_t)n2 + 8 < n1) { n = n2; kd = HMSE_KIND_DELTA; done = true; }
    }
  }
  if (!done) {
    const uint32_t n1 = len_full[k];
    if (n1 == 0xFFFFFFFFu || n1 == 0u) { atomicOr(status, n1 ? 4u : 8u); n = 0; }  // not encodable / never encoded
    else n = n1;
  }
  final_len[k] = n;
  if (kind) kind[k] = kd;
}

__global__ __launch_bounds__(256) void gather_kernel(const uint64_t* __restrict__ cuts, const uint64_t* __restrict__ chunk_ids,
                                                      uint64_t n_sel, const uint64_t* __restrict__ rec_off,
                                                      const uint8_t* __restrict__ recs, const uint8_t* __restrict__ kind,
                                                      const uint64_t* __restrict__ out_off, uint8_t* __restrict__ out,
                                                      uint64_t out_cap, uint32_t* status, const uint64_t* __restrict__ n_dev,
                                                      const uint64_t* __restrict__ out_base_dev) {
  // One chunk per WAVEFRONT (round 4; one per 256-thread workgroup before): a chunk's copy is ~2.7 KiB behind a chain of three dependent
  // metadata loads, so the kernel's time was that chain's latency over the chunks in flight per CU (8); now 32.
  if (n_dev) n_sel = *n_dev;
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t obase = out_base_dev ? *out_base_dev : 0ull;   // captured chain: streams are appended behind the earlier batches'
  // (a wavefront takes every (4 x gridDim.x)-th chunk: the captured chain's grid is sized for the chunks a batch CAN hold)
  for (uint64_t k = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6); k < n_sel; k += (uint64_t)gridDim.x * 4u) {
    const uint64_t c = chunk_ids ? chunk_ids[k] : k;
    const uint32_t len = (uint32_t)(cuts[c + 1] - cuts[c]);
    if (len > 32768u) continue;
    const uint8_t* src = recs + rec_off[k] + rec_slot_off(len, (kind && kind[k] == HMSE_KIND_DELTA) ? 1u : 0u);
    const uint64_t o0 = out_off[k], o1 = out_off[k + 1];
    if (obase + o1 > out_cap) { if (lane == 0) atomicOr(status, 1u); continue; }
    const uint32_t nb = (uint32_t)(o1 - o0);
    uint8_t* dst = out + obase + o0;
    for (uint32_t i = lane * 16; i < nb; i += 64u * 16u) {
      if (i + 16 <= nb) { const uint4 v = *(const uint4*)(src + i); __builtin_memcpy(dst + i, &v, 16); }
      else for (uint32_t b = i; b < nb; b++) dst[b] = src[b];
    }
  }
}

// ---- u64 exclusive scan (three small kernels) ---------------------------------------------------------------
constexpr int SC_NT = 1024;
__global__ __launch_bounds__(SC_NT) void scan_reduce_kernel(const uint64_t* __restrict__ in, uint64_t n, uint64_t* __restrict__ bsum,
                                                            const uint64_t* __restrict__ n_dev) {
  if (n_dev) n = *n_dev;
  __shared__ unsigned long long s[SC_NT / 64];
  const uint64_t i = (uint64_t)blockIdx.x * SC_NT + threadIdx.x;
  unsigned long long v = i < n ? in[i] : 0ull;
  for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
  if (lane_id() == 0) s[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) { unsigned long long tt = 0; for (int w = 0; w < SC_NT / 64; w++) tt += s[w]; bsum[blockIdx.x] = tt; }
}
__global__ __launch_bounds__(SC_NT) void scan_blocks_kernel(uint64_t* bsum, uint64_t nb, uint64_t* total_out, const uint64_t* __restrict__ n_dev) {
  if (n_dev) nb = (*n_dev + SC_NT - 1) / SC_NT;
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
                                                            uint64_t* __restrict__ out, const uint64_t* __restrict__ n_dev) {
  if (n_dev) n = *n_dev;
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
  if (i + 1 == n) out[n] = bsum[blockIdx.x] + wb + inc;   // the total also behind the last element (out has n + 1 entries): with a
                                                           // device-side n the caller's `total` pointer sits at the worst-case index
}

// out[0..n) = exclusive scan of in; *total (device) = sum. in and out may alias.
// (n_dev != nullptr: the element count is read on the device, `n` is the worst case the grids are sized for)
static int exclusive_scan_u64(const uint64_t* in, uint64_t n, uint64_t* out, uint64_t* bsum, uint64_t* total, hipStream_t stream,
                              const uint64_t* n_dev = nullptr) {
  const uint64_t nb = (n + SC_NT - 1) / SC_NT;
  scan_reduce_kernel<<<dim3((uint32_t)nb), dim3(SC_NT), 0, stream>>>(in, n, bsum, n_dev);
  scan_blocks_kernel<<<dim3(1), dim3(SC_NT), 0, stream>>>(bsum, nb, total, n_dev);
  scan_apply_kernel<<<dim3((uint32_t)nb), dim3(SC_NT), 0, stream>>>(in, n, bsum, out, n_dev);
  return hipGetLastError() == hipSuccess ? HMSE_OK : HMSE_EHIP;
}

constexpr int N_WG_B = 256;  // persistent workgroups of the big class (one per CU, global scratch each)

struct Ws {
  uint32_t* counters;  // [c] job count of list c, [32 + c] its cursor
  uint64_t* rec_off; uint64_t* final_len; uint64_t* bsum; uint64_t* rec_total;
  uint32_t* len_full; uint32_t* len_delta; uint32_t* lists; uint64_t list_stride;
  uint8_t* scratch; uint8_t* scratch2; uint8_t* recs; size_t fixed_bytes;
};
static Ws carve(void* ws, uint64_t n_sel) {
  WsCarver w(ws, ~(size_t)0);
  Ws r;
  r.counters = w.take<uint32_t>(N_CTR);
  r.rec_total = w.take<uint64_t>(1);
  r.rec_off = w.take<uint64_t>(n_sel + 1);
  r.final_len = w.take<uint64_t>(n_sel + 1);
  r.bsum = w.take<uint64_t>((n_sel + SC_NT - 1) / SC_NT + 1);
  r.len_full = w.take<uint32_t>(n_sel);
  r.len_delta = w.take<uint32_t>(n_sel);
  r.list_stride = 2 * n_sel;
  r.lists = w.take<uint32_t>(N_LIST * r.list_stride);
  r.scratch = w.take<uint8_t>((size_t)N_WG_B * hmse_align_up(sizeof(Scratch), 256));
  r.scratch2 = w.take<uint8_t>((size_t)512 * hmse_align_up(8 * 32768, 256));   // per match workgroup: match distances / lengths (3 x 32768) + the hand-out list (u32[32768] at byte 98304); per encode workgroup (4096 x 32 KiB, after the match kernels): finished windows of a record that may not be written in place yet
  r.fixed_bytes = w.off;
  r.recs = r.scratch ? (uint8_t*)ws + w.off : nullptr;
  return r;
}

template <bool DICT, int NT, int TCAP, int LCAP, bool LDSM, bool MDG = false, bool MLG = false, bool NOK = false>
static int launch_class(Args a, uint32_t grid, hipStream_t stream) {
  using LY = Layout<NT, TCAP, LCAP, LDSM, MDG, MLG, NOK>;
  // (set once per instantiation; std::call_once: several host threads may enter the library at the same time)
  static std::once_flag attr_once;
  static hipError_t attr_rc = hipSuccess;
  std::call_once(attr_once, [] {
    attr_rc = hipFuncSetAttribute((const void*)l1_deflate_kernel<NT, TCAP, LCAP, LDSM, MDG, MLG, NOK, DICT>, hipFuncAttributeMaxDynamicSharedMemorySize, LY::TOTAL);
  });
  if (attr_rc != hipSuccess) return HMSE_EHIP;
  l1_deflate_kernel<NT, TCAP, LCAP, LDSM, MDG, MLG, NOK, DICT><<<dim3(grid), dim3(NT), LY::TOTAL, stream>>>(a);
  return hipGetLastError() == hipSuccess ? HMSE_OK : HMSE_EHIP;
}

}  // namespace dfl

#ifdef HMSE_DFL_STAMPS
extern "C" int hmse_debug_encode_stamps(unsigned long long* out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(dfl::g_enc_stamps), sizeof(dfl::g_enc_stamps)) != hipSuccess) return HMSE_EHIP;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(dfl::g_enc_stamps), z, sizeof z) != hipSuccess) return HMSE_EHIP; }
  return HMSE_OK;
}
extern "C" int hmse_debug_deflate_stamps(unsigned long long* out96, int reset) {
  if (hipMemcpyFromSymbol(out96, HIP_SYMBOL(dfl::g_dfl_stamps), sizeof(dfl::g_dfl_stamps)) != hipSuccess) return HMSE_EHIP;
  if (reset) { unsigned long long z[144] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(dfl::g_dfl_stamps), z, sizeof z) != hipSuccess) return HMSE_EHIP; }
  return HMSE_OK;
}
#endif

extern "C" uint64_t hmse_l1_deflate_record_bytes(uint32_t chunk_len) { return chunk_len <= 32768u ? dfl::rec_size(chunk_len, false) : 0u; }
extern "C" uint64_t hmse_l1_deflate_record_bytes_dict(uint32_t chunk_len) { return chunk_len <= 32768u ? dfl::rec_size(chunk_len, true) : 0u; }

// fixed part only; the caller adds the record area: sum over chunks of hmse_l1_deflate_record_bytes[_dict](len)
size_t hmse_l1_deflate_workspace_bytes_impl(uint64_t n_sel, const hmse_cfg*) { return dfl::carve(nullptr, n_sel).fixed_bytes; }

static int deflate_impl(const uint8_t* data, uint64_t n, const uint64_t* cuts, const uint64_t* chunk_ids,
                        const int64_t* base, uint64_t n_sel, const uint64_t* n_dev, const uint64_t* out_base_dev, const hmse_cfg* cfg, uint32_t flags,
                        uint8_t* out, uint64_t out_cap, uint64_t* out_off, uint8_t* kind, uint32_t* status, void* ws, size_t ws_bytes,
                        void* stream_);

extern "C" int hmse_l1_deflate_ex(const uint8_t* data, uint64_t n, const uint64_t* cuts, const uint64_t* chunk_ids,
                                  const int64_t* base, uint64_t n_sel, const hmse_cfg* cfg, uint32_t flags, uint8_t* out,
                                  uint64_t out_cap, uint64_t* out_off, uint8_t* kind, uint32_t* status, void* ws, size_t ws_bytes,
                                  void* stream_) {
  return deflate_impl(data, n, cuts, chunk_ids, base, n_sel, nullptr, nullptr, cfg, flags, out, out_cap, out_off, kind, status, ws, ws_bytes, stream_);
}

// captured chain: the selection's size is *n_sel_dev (<= cap_sel, which sizes grids and workspace); bases are chunk ids; the total
// stream size lands in out_off[cap_sel]
int hmse_l1_deflate_dyn(const uint8_t* data, uint64_t n_cap, const uint64_t* cuts_all, const uint64_t* sel_ids, const int64_t* sel_base,
                        const uint64_t* n_sel_dev, uint64_t cap_sel, const uint64_t* out_base_dev, const hmse_cfg* cfg, uint8_t* out, uint64_t out_cap,
                        uint64_t* out_off, uint8_t* kind, uint32_t* status, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (!n_sel_dev || cap_sel == 0) return HMSE_EINVAL;
  return deflate_impl(data, n_cap, cuts_all, sel_ids, sel_base, cap_sel, n_sel_dev, out_base_dev, cfg, HMSE_DEFLATE_BASE_IS_CHUNK_ID, out, out_cap, out_off,
                      kind, status, ws, ws_bytes, (void*)stream);
}

// n_dev == nullptr: n_sel is the selection's size.  n_dev != nullptr: n_sel is the worst case (grids, workspace carve), the real
// size is read on the device by every kernel that needs it.
static int deflate_impl(const uint8_t* data, uint64_t n, const uint64_t* cuts, const uint64_t* chunk_ids,
                        const int64_t* base, uint64_t n_sel, const uint64_t* n_dev, const uint64_t* out_base_dev, const hmse_cfg* cfg, uint32_t flags,
                        uint8_t* out, uint64_t out_cap, uint64_t* out_off, uint8_t* kind, uint32_t* status, void* ws, size_t ws_bytes,
                        void* stream_) {
  using namespace dfl;
  if (hmse_cfg_validate_impl(cfg) != 0) return HMSE_EINVAL;
  if (flags & ~(uint32_t)HMSE_DEFLATE_BASE_IS_CHUNK_ID) return HMSE_EINVAL;
  const uint32_t a_base_is_chunk = (flags & HMSE_DEFLATE_BASE_IS_CHUNK_ID) ? 1u : 0u;
  if (!out_off || !status) return HMSE_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  (void)hipGetLastError();  // drop stale errors of earlier runtime calls made by the host process
  HMSE_FILL(status, 0, sizeof(uint32_t), stream);
  if (n_sel == 0) { HMSE_FILL(out_off, 0, sizeof(uint64_t), stream); return HMSE_OK; }
  if (!data || !cuts || !out || !kind) return HMSE_EINVAL;
  if (n_sel > 0x3FFFFFFFull) return HMSE_EINVAL;
  Ws w = carve(ws, n_sel);
  if (!ws || ws_bytes < w.fixed_bytes) return HMSE_ENOSPC;
  // record area = whatever follows the fixed part; a job whose record does not fit sets status bit 1
  // (needed: sum over selected chunks of rec_size(len, has a base) ~ 4*len + 1.4 KiB, + len + 21 where a base exists)
  const uint64_t avail = ws_bytes - w.fixed_bytes;
  HMSE_FILL(w.counters, 0, N_CTR * sizeof(uint32_t), stream);
  HMSE_FILL(w.len_full, 0, n_sel * sizeof(uint32_t), stream);
  HMSE_FILL(w.len_delta, 0, n_sel * sizeof(uint32_t), stream);
  const uint32_t blocks = (uint32_t)((n_sel + 255) / 256);
  rec_size_kernel<<<dim3(blocks), dim3(256), 0, stream>>>(cuts, chunk_ids, base, n_sel, w.rec_off, n_dev);
  HMSE_LAUNCH_CHECK();
  if (exclusive_scan_u64(w.rec_off, n_sel, w.rec_off, w.bsum, w.rec_total, stream, n_dev) != HMSE_OK) return HMSE_EHIP;
  classify_kernel<<<dim3((uint32_t)((n_sel + LIST_NT - 1) / LIST_NT)), dim3(LIST_NT), 0, stream>>>(cuts, chunk_ids, base, n_sel, a_base_is_chunk, w.lists, w.list_stride, w.counters, n_dev);
  HMSE_LAUNCH_CHECK();
  Args a;
  a.data = data; a.n = n; a.cuts = cuts; a.chunk_ids = chunk_ids; a.base = base; a.n_sel = n_sel;
  a.depth = hmse_deflate_depth(cfg);
  a.base_is_chunk = a_base_is_chunk;
  a.rec_off = w.rec_off; a.recs = w.recs; a.len_full = w.len_full; a.len_delta = w.len_delta;
  a.rec_cap = avail; a.status = status;
  a.scratch = w.scratch; a.scratch_stride = hmse_align_up(sizeof(Scratch), 256);
  a.scratch2 = w.scratch2; a.scratch2_stride = hmse_align_up(8 * 32768, 256);
  // persistent grids: small class 2 workgroups per CU, medium 1 per CU, big class a handful
  const uint64_t max_jobs = 2 * n_sel;  // (upper bound of any list)
  // big windows first (few, long jobs), then the LDS classes
  auto sel = [&](int c) { a.jobs = w.lists + (size_t)c * w.list_stride; a.n_jobs = w.counters + c; a.counter = w.counters + 32 + c; };
  // HMSE_DFL_DEBUG_SYNC=1 (diagnostics): wait after every launch and say which one returned — localises a kernel that does not finish
  static const bool dbg_sync = getenv("HMSE_DFL_DEBUG_SYNC") != nullptr;
  auto dbg = [&](int slot) { if (dbg_sync) { const hipError_t e = hipStreamSynchronize(stream); fprintf(stderr, "[hmse_l1_deflate] launch of slot %d finished (%d)\n", slot, (int)e); fflush(stderr); } };
  a.prof_ctr = g_hmse_prof ? g_hmse_prof_ctr : nullptr; a.prof_slot = 0;
  // Order (rule 7): dictionary jobs -> their DELTA encodes -> redo_kernel (chunks whose delta is no quick accept join the plain
  // lists) -> plain jobs -> FULL encodes.  Profile slots: plain 8..13, dictionary 18..23, encode FULL 14 / 15, encode DELTA 30 / 31.
  // HMSE_ENC_FORCE_SPILL=1 (tests): every FULL record takes the encode kernel's spill path (finished windows via the global scratch)
  static const bool enc_force_spill = getenv("HMSE_ENC_FORCE_SPILL") != nullptr;
#define HMSE_DFL_LAUNCH(DICT_, LIST, SLOT, GRID, ...)                                                               \
  sel(LIST); a.prof_slot = (SLOT);                                                                                   \
  PROF_BEGIN(SLOT, stream);                                                                                          \
  if (launch_class<DICT_, __VA_ARGS__>(a, (uint32_t)(max_jobs < (uint64_t)(GRID) ? max_jobs : (uint64_t)(GRID)), stream) != HMSE_OK) return HMSE_EHIP; \
  PROF_END(SLOT, stream); dbg(SLOT);
#define HMSE_DFL_ENCODE(LIST1, SLOT1, LIST2, SLOT2)                                                                  \
  sel(LIST1); a.prof_slot = (SLOT1);                                                                                 \
  PROF_BEGIN(SLOT1, stream);                                                                                         \
  l1_encode_kernel<0, 12288><<<dim3((uint32_t)(max_jobs < 4096 ? max_jobs : 4096)), dim3(128), EncLayout<0, 12288>::TOTAL, stream>>>(a, enc_force_spill);           \
  PROF_END(SLOT1, stream); dbg(SLOT1);                                                                               \
  sel(LIST2); a.prof_slot = (SLOT2);                                                                                 \
  PROF_BEGIN(SLOT2, stream);                                                                                         \
  l1_encode_kernel<12288, 32768><<<dim3((uint32_t)(max_jobs < 3328 ? max_jobs : 3328)), dim3(128), EncLayout<12288, 32768>::TOTAL, stream>>>(a, enc_force_spill);   \
  PROF_END(SLOT2, stream); dbg(SLOT2);                                                                               \
  HMSE_LAUNCH_CHECK();
  if (base) {
    // big windows first (few, long jobs), then the LDS classes
    HMSE_DFL_LAUNCH(true, (int)dict_list(3), 18 + 3, N_WG_B, NT_B, TCAP_B, LCAP_B, false)
    HMSE_DFL_LAUNCH(true, (int)dict_list(2), 18 + 2, 256, NT_M, TCAP_SG3, TCAP_SG3, true, true, true, true)
    HMSE_DFL_LAUNCH(true, (int)dict_list(8), 18 + 1, 512, NT_S, TCAP_SG2, TCAP_SG2, true, true, true, true)
    HMSE_DFL_LAUNCH(true, (int)dict_list(5), 18 + 5, 512, NT_S, TCAP_SG, TCAP_SG, true, true, true)
    HMSE_DFL_LAUNCH(true, (int)dict_list(4), 18 + 4, 512, NT_S, TCAP_S2, TCAP_S2, true, true)
    HMSE_DFL_LAUNCH(true, (int)dict_list(0), 18 + 0, 512, NT_S, TCAP_S, TCAP_S, true)
    HMSE_DFL_ENCODE(15, 30, 16, 31)
    redo_kernel<<<dim3((uint32_t)((n_sel + LIST_NT - 1) / LIST_NT)), dim3(LIST_NT), 0, stream>>>(cuts, chunk_ids, base, n_sel, w.len_delta, cfg->delta_max_ratio_pct, w.lists, w.list_stride,
                                                        w.counters, n_dev);
    HMSE_LAUNCH_CHECK();
  }
  HMSE_DFL_LAUNCH(false, 3, 8 + 3, N_WG_B, NT_B, TCAP_B, LCAP_B, false)
  HMSE_DFL_LAUNCH(false, 2, 8 + 2, 256, NT_M, TCAP_SG3, TCAP_SG3, true, true, true, true)
  HMSE_DFL_LAUNCH(false, 8, 8 + 1, 512, NT_S, TCAP_SG2, TCAP_SG2, true, true, true, true)
  HMSE_DFL_LAUNCH(false, 5, 8 + 5, 512, NT_S, TCAP_SG, TCAP_SG, true, true, true)
  HMSE_DFL_LAUNCH(false, 4, 8 + 4, 512, NT_S, TCAP_S2, TCAP_S2, true, true)
  HMSE_DFL_LAUNCH(false, 0, 8 + 0, 512, NT_S, TCAP_S, TCAP_S, true)
  HMSE_DFL_ENCODE(6, 14, 7, 15)
#undef HMSE_DFL_LAUNCH
#undef HMSE_DFL_ENCODE
  decide_kernel<<<dim3(blocks), dim3(256), 0, stream>>>(cuts, chunk_ids, base, n_sel, w.len_full, w.len_delta,
                                                        cfg->delta_max_ratio_pct, w.final_len, kind, status, n_dev);
  HMSE_LAUNCH_CHECK();
  if (exclusive_scan_u64(w.final_len, n_sel, out_off, w.bsum, out_off + n_sel, stream, n_dev) != HMSE_OK) return HMSE_EHIP;
  gather_kernel<<<dim3((uint32_t)((n_sel + 3) / 4 < 16384 ? (n_sel + 3) / 4 : 16384)), dim3(256), 0, stream>>>(cuts, chunk_ids, n_sel, w.rec_off, w.recs, kind, out_off, out,
                                                                out_cap, status, n_dev, out_base_dev);
  HMSE_LAUNCH_CHECK();
  return HMSE_OK;
}

extern "C" int hmse_l1_deflate(const uint8_t* data, uint64_t n, const uint64_t* cuts, const uint64_t* chunk_ids,
                               const int64_t* base, uint64_t n_sel, const hmse_cfg* cfg, uint8_t* out, uint64_t out_cap,
                               uint64_t* out_off, uint8_t* kind, uint32_t* status, void* ws, size_t ws_bytes, void* stream_) {
  return hmse_l1_deflate_ex(data, n, cuts, chunk_ids, base, n_sel, cfg, 0u, out, out_cap, out_off, kind, status, ws, ws_bytes, stream_);
}
