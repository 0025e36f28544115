// l1_inflate.hip — read path: raw DEFLATE decode of every stored chunk on the GPU.
//
// Replaces mz_inflateInit2(&s, 15) + mz_inflate(&s, MZ_FINISH) (README.md:2397-2400) and the three-branch
// reconstruction of README.md:1621-1675 / 2191-2198:  FULL -> inflate;  DELTA -> inflate with the base chunk's
// raw bytes as preset dictionary;  POINTER -> the target chunk (hmse_read_assemble).
//
// DEFLATE decoding is serial inside a stream, so the parallel axis is streams: ONE STREAM PER WAVEFRONT,
// persistent wavefronts pulling streams in index order.  Inside a stream the 64 lanes cooperate:
//   * canonical Huffman decode in one step per symbol: lane l tests "do the next l bits form a code of length
//     l" against first-code/count per length; a ballot picks the (unique) matching length;
//   * code-length tables are built with wave ballots/prefix counts;
//   * match copies run 64 bytes per step; a self-overlapping match (dist < len) is periodic, so lane i reads
//     byte (i mod dist) of the period that precedes the match: no read-after-write inside a match.
// A DELTA stream needs its base chunk decoded first.  Streams are handed out in index order and base[k] < k,
// so the base's wavefront is already running or done: the consumer polls a per-chunk flag (agent-scope
// release/acquire as in the guide's hand-off recipe) — forward progress is guaranteed.
// The bit reader keeps a 64-bit window refilled from global memory; the output is re-read (match sources)
// with agent-scope (L1-bypassing) loads after the wave's own stores have drained.
#include "common.h"
#include <atomic>

namespace ifl {

constexpr int NT = 256;  // 4 streams per workgroup

struct WaveTables {
  uint16_t lsym[288];   // lit/len symbols sorted by (length, symbol)
  uint16_t dsym[32];
  uint8_t lens[320];    // code lengths while a dynamic header is read
};

struct Args {
  const uint8_t* streams; uint64_t streams_bytes; const uint64_t* stream_off; const uint32_t* stream_len;
  const uint8_t* kind; const int64_t* base; uint32_t n_sel;
  const uint64_t* raw_off; uint8_t* raw_out; uint64_t raw_cap;
  uint32_t* status; uint8_t* ok; uint32_t* done; uint32_t* counter;
  uint32_t dflags;  // diagnostics only (HMSE_DIAG build): 1 = skip literal stores, 2 = skip match stores, 4 = skip match loads
  uint32_t* trace;  // diagnostics only (tools/inflate_debug.py): host-visible progress words, 8 per wavefront
};

#ifdef HMSE_DIAG
#define IFL_TRACE(slot, val) do { if (a.trace && lane == 0) __hip_atomic_store(&a.trace[(blockIdx.x * (NT / 64) + wave) * 8 + (slot)], (uint32_t)(val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } while (0)
#else
#define IFL_TRACE(slot, val) do { } while (0)
#endif

// CONVERGENCE RULE of this file: the compiler does not promise that lanes which skipped an `if (lane == ...)` block
// wait for the others before the next wave-level operation (readfirstlane, ballot, readlane) — with such a block at
// the tail of a loop body it let lanes 1..63 run into the next iteration, where "first active lane" was no longer
// lane 0 (measured: an endless loop on the GPU).  So no lane-dependent branch sits at a loop tail here: loops have
// wave-uniform trip counts with the lane test inside, and work for "one lane" is done by all lanes storing the same
// value to the same address (the hardware merges them).
// Everything that steers control flow is wave-uniform and is kept in SGPRs: values that arrive through a vector
// load (same address in every lane) are made scalar with readfirstlane, so the decoder's loops are scalar branches.
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uni64(uint64_t v) { return ((uint64_t)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v); }
__device__ __forceinline__ uint64_t uload64(const uint64_t* p) { return uni64(*p); }

// bit reader over streams[p, end): 64-bit window, LSB first (RFC 1951 §3.1.1); all state scalar
struct Bits {
  const uint8_t* s; uint64_t p, end; uint64_t acc; uint32_t n;
  __device__ __forceinline__ void refill() {
    if (p + 8 <= end) {  // one unaligned 8-byte load at a uniform address
      uint64_t w;
      __builtin_memcpy(&w, s + p, 8);
      acc |= uni64(w) << n;
      const uint32_t adv = (63u - n) >> 3;
      p += adv; n += adv * 8u;
      return;
    }
    while (n <= 56) {  // last bytes of the stream; zeros behind it (the final length check catches their use)
      uint32_t b = 0;
      if (p < end) b = uni(s[p]);
      p++;
      acc |= (uint64_t)b << n; n += 8;
    }
  }
  __device__ __forceinline__ uint32_t peek(uint32_t k) const { return (uint32_t)acc & ((1u << k) - 1u); }  // k <= 16
  __device__ __forceinline__ void drop(uint32_t k) { acc >>= k; n -= k; }
  __device__ __forceinline__ uint32_t take(uint32_t k) { if (n < k) refill(); const uint32_t v = peek(k); drop(k); return v; }
};

// Per-lane view of a canonical code: lane l (1..15) owns code length l.
struct LaneCode { uint32_t count, first, offs; };

// Build the canonical decode tables from code lengths lens[0..n): per length the count, first code and offset into
// `sym`, which receives the symbols sorted by (length, symbol).  Returns false for a set stock zlib rejects:
// over-subscribed, or incomplete unless its longest code has length 1 (never allowed for the code-length code).
__device__ bool build_table(const uint8_t* lens, uint32_t n, uint16_t* sym, bool is_codes, LaneCode& lc) {
  const uint32_t lane = lane_id();
  uint32_t mycnt = 0;
  for (uint32_t b = 0; b < n; b += 64) {
    const uint32_t s = b + lane;
    const uint32_t l = s < n ? lens[s] : 0u;
#pragma nounroll
    for (uint32_t k = 1; k < 16; k++) {
      const uint32_t c = (uint32_t)__builtin_popcountll(__ballot(l == k));
      mycnt += lane == k ? c : 0u;
    }
  }
  uint32_t code = 0, off = 0, prevc = 0, maxl = 0, myfirst = 0, myoff = 0;
  int32_t left = 1;
  bool over = false;
#pragma nounroll
  for (uint32_t k = 1; k < 16; k++) {
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)mycnt, (int)k);
    code = (code + prevc) << 1;
    myfirst = lane == k ? code : myfirst;
    myoff = lane == k ? off : myoff;
    off += c; prevc = c;
    left = (left << 1) - (int32_t)c;
    if (left < 0) over = true;
    if (c) maxl = k;
  }
  const bool mine = lane >= 1 && lane <= 15;
  lc.count = mine ? mycnt : 0u; lc.first = myfirst; lc.offs = myoff;
  if (over || (maxl != 0 && left > 0 && (is_codes || maxl != 1))) return false;
  // symbols of equal length keep index order: rank inside the length by ballots
  uint32_t myrun = myoff;
  for (uint32_t b = 0; b < n; b += 64) {
    const uint32_t s = b + lane;
    const uint32_t l = s < n ? lens[s] : 0u;
#pragma nounroll
    for (uint32_t k = 1; k < 16; k++) {
      const uint64_t m = __ballot(l == k);
      if (m == 0) continue;
      const uint32_t at = (uint32_t)__builtin_amdgcn_readlane((int)myrun, (int)k);
      if (l == k) sym[at + (uint32_t)__builtin_popcountll(m & lanemask_lt())] = (uint16_t)s;
      myrun += lane == k ? (uint32_t)__builtin_popcountll(m) : 0u;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  return true;
}

// One symbol in one step: lane l tests whether the next l bits are a code of length l (canonical codes: the l-bit
// value minus the first code of that length indexes the symbols of that length); the shortest hit is the code.
// Returns 0xFFFF on an undefined code.
__device__ __forceinline__ uint32_t decode_sym(Bits& br, const LaneCode& lc, const uint16_t* sym) {
  if (br.n < 15) br.refill();
  const uint32_t l = lane_id() & 15u;
  const uint32_t rev = __builtin_bitreverse32(br.peek(15));                     // first stream bit -> bit 31
  const uint32_t code = l ? rev >> (32u - l) : 0u;                              // first l bits, MSB-first value
  const uint32_t rel = code - lc.first;
  const uint64_t m = __ballot(rel < lc.count);                                  // count is 0 outside lanes 1..15
  if (m == 0) return 0xFFFFu;
  const uint32_t len = (uint32_t)__builtin_ctzll(m);
  const uint32_t idx = (uint32_t)__builtin_amdgcn_readlane((int)(lc.offs + rel), (int)len);
  br.drop(len);
  return uni(sym[idx]);
}

__global__ __launch_bounds__(NT, 8) void l1_inflate_kernel(Args a) {
  __shared__ WaveTables s_tab[NT / 64];
  const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
  WaveTables& T = s_tab[wave];
  for (;;) {
    // one pull per wavefront: every lane takes part (lane 0 adds 1, the others 0; the compiler folds this into a
    // single atomic) so that no lane-dependent branch precedes the readfirstlane
    const uint32_t k = uni(atomicAdd(a.counter, lane == 0 ? 1u : 0u));
    IFL_TRACE(0, 1); IFL_TRACE(1, k);
    if (k >= a.n_sel) break;
    const uint64_t o0 = uload64(a.raw_off + k), o1 = uload64(a.raw_off + k + 1);
    const uint32_t L = (uint32_t)(o1 - o0);
    uint8_t* const out = a.raw_out + o0;
    bool bad = o1 < o0 || o1 > a.raw_cap;
    // dictionary = base chunk's raw bytes (DELTA only): wait until its wavefront has published it.  Streams are handed
    // out in index order and base < k, so that wavefront is running or done; the wait is bounded all the same
    // (a wavefront must always reach its exit) and running out of polls is reported as status bit 1.
    const uint8_t* dict = nullptr; uint32_t Dl = 0;
    if (!bad && uni(a.kind[k]) == HMSE_KIND_DELTA) {
      const int64_t b = a.base ? (int64_t)uload64((const uint64_t*)a.base + k) : -1;
      if (b < 0 || (uint64_t)b >= k) bad = true;
      else {
        IFL_TRACE(0, 2);
        uint32_t polls = 0, st;
        while ((st = uni(__hip_atomic_load(&a.done[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) == 0u && ++polls < (1u << 21))
          __builtin_amdgcn_s_sleep(16);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (st == 0u) { bad = true; atomicOr(a.status, 2u); }
        if (st == 2u) bad = true;  // the base failed to decode
        const uint64_t b0 = uload64(a.raw_off + b), b1 = uload64(a.raw_off + b + 1);
        uint64_t dl = b1 - b0;
        dict = a.raw_out + b0;
        if (dl > 32768) { dict += dl - 32768; dl = 32768; }
        Dl = (uint32_t)dl;
      }
    }
    const uint64_t s0 = uload64(a.stream_off + k);
    const uint64_t s1 = a.stream_len ? s0 + uni(a.stream_len[k]) : uload64(a.stream_off + k + 1);
    if (s1 < s0 || s1 > a.streams_bytes) bad = true;
    Bits br; br.s = a.streams; br.p = s0; br.end = bad ? s0 : s1; br.acc = 0; br.n = 0;
    uint32_t pos = 0;      // bytes decoded, including literals still pending in `pend`
    uint32_t npend = 0;    // literals not yet stored: lane i of `pend` holds byte pos - npend + i
    uint32_t pend = 0;
    bool last = false;
    // every block costs >= 3 bits and every symbol >= 1 bit: a step budget no valid stream can exceed, so that a
    // wavefront reaches its exit whatever the bytes are
    uint64_t budget = 8ull * (br.end - s0) + 64;
    LaneCode LL{0, 0, 0}, LD{0, 0, 0};
    while (!bad && !last) {
      if (budget-- == 0) { bad = true; break; }
      IFL_TRACE(0, 3); IFL_TRACE(2, pos);
      last = br.take(1) != 0;
      const uint32_t type = br.take(2);
      if (type == 0) {  // stored
        br.drop(br.n & 7u);
        const uint32_t len = br.take(16), nlen = br.take(16);
        if ((len ^ nlen) != 0xFFFFu || pos + len > L) { bad = true; break; }
        const uint64_t src = br.p - (br.n >> 3);  // whole bytes still in the window come first
        if (src + len > br.end) { bad = true; break; }
        if (npend) { if (lane < npend) out[pos - npend + lane] = (uint8_t)pend; npend = 0; }
        for (uint32_t i0 = 0; i0 < len; i0 += 64) {
          const uint32_t i = i0 + lane;
          if (i < len) out[pos + i] = a.streams[src + i];
        }
        pos += len;
        br.p = src + len; br.acc = 0; br.n = 0;
        continue;
      }
      if (type == 3) { bad = true; break; }
      if (type == 1) {  // fixed codes
        for (uint32_t s0 = 0; s0 < 320; s0 += 64) {
          const uint32_t s = s0 + lane;
          if (s < 288) T.lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        build_table(T.lens, 288, T.lsym, false, LL);
        if (lane < 32) T.lens[lane] = 5;   // 32 five-bit distance codes; 30 and 31 are rejected where they are used
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        build_table(T.lens, 32, T.dsym, false, LD);
      } else {  // dynamic codes
        const uint32_t nlit = br.take(5) + 257, ndist = br.take(5) + 1, ncl = br.take(4) + 4;
        if (nlit > 286 || ndist > 30) { bad = true; break; }
        if (lane < 19) T.lens[lane] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        for (uint32_t i = 0; i < ncl; i++) {
          const uint32_t v = br.take(3);
          // order of the code-length code lengths (RFC 1951 §3.2.7): 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
          const uint32_t o = i < 3 ? 16 + i : i == 3 ? 0 : (i & 1) ? (19 - i) >> 1 : 6 + (i >> 1);
          T.lens[o] = (uint8_t)v;  // every lane, same address and value
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        LaneCode LC;
        if (!build_table(T.lens, 19, T.dsym, true, LC)) { bad = true; break; }  // code-length code borrows dsym
        // lengths are decoded into lens[LB..): lens[0..19) still feeds nothing, dsym is read by decode_sym
        uint32_t i = 0, prev = 0;
        const uint32_t tot = nlit + ndist;
        while (i < tot) {
          const uint32_t s = decode_sym(br, LC, T.dsym);
          if (s > 18) { bad = true; break; }
          uint32_t rep = 1, val = s;
          if (s == 16) { if (i == 0) { bad = true; break; } rep = 3 + br.take(2); val = prev; }
          else if (s == 17) { rep = 3 + br.take(3); val = 0; }
          else if (s == 18) { rep = 11 + br.take(7); val = 0; }
          if (i + rep > tot) { bad = true; break; }
          for (uint32_t j0 = 0; j0 < rep; j0 += 64) {
            const uint32_t j = j0 + lane;
            if (j < rep) T.lens[i + j] = (uint8_t)val;
          }
          i += rep; prev = val;
        }
        if (bad) break;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        if (uni(T.lens[256]) == 0) { bad = true; break; }  // no end-of-block code
        if (!build_table(T.lens, nlit, T.lsym, false, LL)) { bad = true; break; }
        if (!build_table(T.lens + nlit, ndist, T.dsym, false, LD)) { bad = true; break; }
      }
      IFL_TRACE(0, 6);
      // ---- symbols of this block ----
      for (;;) {
        if (budget-- == 0) { bad = true; break; }
        const uint32_t s = decode_sym(br, LL, T.lsym);
        if (s < 256) {  // literal: parked in lane npend of `pend`, stored 64 at a time
          if (pos >= L) { bad = true; break; }
          pend = lane == npend ? s : pend;
          pos++; npend++;
          if (npend == 64) { out[pos - 64 + lane] = (uint8_t)pend; npend = 0; }
          continue;
        }
        if (s == 256) break;
        if (s > 285) { bad = true; break; }  // 286, 287 and the undefined-code marker 0xFFFF
        // length and distance (RFC 1951 §3.2.5)
        const uint32_t lc = s - 257;
        uint32_t len;
        if (lc < 8) len = 3 + lc;
        else if (lc == 28) len = 258;
        else { const uint32_t e = (lc - 4) >> 2; len = 3 + ((4 + (lc & 3)) << e) + br.take(e); }
        const uint32_t ds = decode_sym(br, LD, T.dsym);
        if (ds > 29) { bad = true; break; }
        uint32_t dist;
        if (ds < 4) dist = 1 + ds;
        else { const uint32_t e = (ds >> 1) - 1; dist = 1 + ((2 + (ds & 1)) << e) + br.take(e); }
        if (pos + len > L || dist > pos + Dl) { bad = true; break; }
        if (npend) { if (lane < npend) out[pos - npend + lane] = (uint8_t)pend; npend = 0; }
        // the wave's own earlier stores must have landed before they are read back
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // every byte of a match equals out[pos - dist + (i mod dist)], which lies before the match: one pass,
        // no read-after-write inside the match even when it overlaps itself (dist < len)
        for (uint32_t i0 = 0; i0 < len; i0 += 64) {
          const uint32_t i = i0 + lane;
          if (i < len) {
            const uint32_t j = dist >= len ? i : i % dist;
            const int64_t sp = (int64_t)pos - dist + j;  // < 0: inside the dictionary
            const uint8_t* srcp = sp >= 0 ? out + sp : dict + (int64_t)Dl + sp;
            out[pos + i] = __hip_atomic_load(srcp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        pos += len;
      }
    }
    if (npend && !bad) { if (lane < npend) out[pos - npend + lane] = (uint8_t)pend; }
    // the storage contract: exactly the recorded raw length, and the final block ends in the stream's last byte
    if (!bad && (pos != L || br.p - (br.n >> 3) != br.end)) bad = true;
    IFL_TRACE(0, 7); IFL_TRACE(2, pos);
    // publish: every lane's stores drained, agent-scope release, then the flag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    // every lane stores the same words (see the convergence rule above)
    __hip_atomic_store(&a.done[k], bad ? 2u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.ok) a.ok[k] = bad ? 0 : 1;
    if (bad) atomicOr(a.status, 1u);
    IFL_TRACE(0, 8);
  }
  IFL_TRACE(0, 9);
}

}  // namespace ifl

// =====================================================================================================================
// ONE STREAM PER LANE (the wide kernel, taken when a call carries enough streams to fill the chip's lanes).
//
// The wavefront-per-stream kernel above keeps 63 lanes waiting on one scalar bit reader: per token it issues ~100
// scalar and ~45 vector instructions, and the CU's single scalar unit is what bounds it (profiles/r2: 2.25 scalar
// per vector instruction, VALU half idle).  Here every lane is a complete, independent DEFLATE decoder: 64 streams per
// wavefront, no cross-lane operation in the data path at all.  What makes that fit the machine:
//   * canonical Huffman decode WITHOUT a length loop or a lookup table: for each code length l the table build leaves a
//     packed word  limit_l << 16 | info  in a REGISTER (limit_l = left-aligned end of the codes of length <= l).  The
//     next 15 stream bits, bit-reversed, are compared with all 15 words (v_cmp + v_cndmask each, fully unrolled, static
//     register indices): the last word not above the bits carries the code's length, the first code of that length and
//     the offset of its symbols.  One LDS byte read then gives the symbol.
//   * per lane, LDS holds ONLY the 288 sorted literal/length symbol bytes (bit 8 of a symbol comes from comparing its
//     index with the count of literals of that length), in lane-interleaved words (word w of lane l at w*64+l: a byte
//     read is conflict-free whatever the indices): 18 KiB per wavefront, EIGHT wavefronts per CU (two per SIMD),
//     131 072 streams in flight on the chip.  Everything else is in registers, none of it indexed by a lane: the 30
//     distance symbols and the per-length counts/offsets in packed 64-bit words, the code lengths of a header in a
//     40-register FIFO (a queue only ever moves by one);
//   * a state machine per lane, one micro-step per loop trip: a short-code literal (slot A) plus one token, or up to 14
//     bytes of a match.  All memory operations of a trip sit in ONE cluster at its end (step 5) whose single wait is for
//     what the previous trip issued: the stream is read 16 bytes at a time into a register window, a match piece is one
//     16-byte load appended a trip later, output leaves through a 16-byte register buffer as one store per ~12 bytes;
//     a self-overlapping match doubles its copy distance (a multiple of dist) until the piece fits;
//   * block headers (code-length code, length decode, table build: ~15 k instructions per lane) would run with one or
//     two lanes active if each lane did them when it got there.  Lanes that need a header WAIT until 16 of them do (or
//     nothing else can run), then go through it together, reading the header from 288 stream bytes staged in the
//     (not yet needed) symbol area;
//   * records are pulled in two passes, those without a dictionary first: a DELTA record then finds its base decoded
//     (in index order it sat right behind it and its lane idled for the base's whole decode).  A base still in flight
//     (chains of DELTAs, the seam of the passes) is polled every eighth trip, never spun on: it may be another lane of
//     this very wavefront.
// Every wait is bounded (symbol budget per stream, poll budget per base, bytes per copy), so every lane reaches DONE.
namespace ifl {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int LW = 72;                          // LDS words per lane: the sorted literal/length symbols, one byte each
constexpr int W_LSYM = 0;
constexpr uint32_t ST_WAIT = 0, ST_DEC = 1, ST_FIN = 2, ST_DONE = 3;
#ifndef HMSE_HDR_MIN
#define HMSE_HDR_MIN 16
#endif
constexpr uint32_t HDR_MIN = HMSE_HDR_MIN;                // lanes that must be waiting before a header round is run
constexpr uint32_t POLL_MAX = 1u << 21;

__device__ __forceinline__ uint64_t load8_at(const uint8_t* base, uint64_t off, uint64_t total) {
  uint64_t w = 0;
  if (off + 8 <= total) __builtin_memcpy(&w, base + off, 8);
  else for (uint32_t j = 0; j < 8; j++) if (off + j < total) w |= (uint64_t)base[off + j] << (8 * j);
  return w;
}

// length of the code that the left-aligned 15-bit value X starts with, and where its symbols are:
// T[j] = limit_j << 16 | info, ascending in j; the last entry with limit_j <= X wins (T[0] has limit 0)
#define IFL_SCAN(X, T, NL, sel)                                        \
  do {                                                                 \
    const uint32_t xk_ = ((X) << 16) | 0xFFFFu;                        \
    sel = T[0];                                                        \
    _Pragma("unroll") for (int j_ = 1; j_ <= NL; j_++) sel = xk_ >= T[j_] ? T[j_] : sel; \
  } while (0)

#ifdef HMSE_DIAG
// cycle shares of the trip's steps, summed over wavefronts into a.trace as u64[8] (tools/inflate_lanes_cycles.py):
// trips, header rounds, cycles in publish / header / poll / decode / memory cluster, lane-trips that decoded a token
#define LK_T0() uint64_t t_ = __builtin_readcyclecounter()
#define LK_LAP(i) do { const uint64_t u_ = __builtin_readcyclecounter(); cyc[i] += u_ - t_; t_ = u_; } while (0)
#define LK_SUB(i) do { const uint64_t u_ = __builtin_readcyclecounter(); cyc[i] += u_ - ts_; ts_ = u_; } while (0)
#else
#define LK_T0() do { } while (0)
#define LK_LAP(i) do { } while (0)
#define LK_SUB(i) do { } while (0)
#endif

__global__ __launch_bounds__(64, 2) void l1_inflate_lanes_kernel(Args a) {
  __shared__ uint32_t lds[LW * 64];
  const uint32_t lane = threadIdx.x;
  uint32_t* const my = lds + lane;                                    // word w at my[w * 64]
  uint8_t* const myb = (uint8_t*)lds + lane * 4;                      // byte B at myb[(B >> 2) * 256 + (B & 3)]
#define LB(B) myb[(((uint32_t)(B)) >> 2) * 256u + (((uint32_t)(B)) & 3u)]

  uint64_t ds0 = 0, ds1 = 0, ds2 = 0;                                 // distance symbols sorted by (length, symbol), 5 bits each
  uint32_t P[16], Q[16], PD[16];                                      // decode words of the current block (see IFL_SCAN)
#pragma unroll
  for (int j = 0; j < 16; j++) { P[j] = 0; Q[j] = 0; PD[j] = 0; }
  uint32_t st = ST_WAIT;
  bool pass2 = false;
  bool need_pull = true;
  // lane flags as 0/1 words in VGPRs: as bools they live as lane masks in SGPRs, and every update in divergent code costs
  // three scalar mask operations
  uint32_t bad = 0, last = 0, eob = 0, blocked = 0, stored = 0, force = 0;
  uint32_t k = 0, L = 0, pos = 0, Dl = 0, budget = 0, polls = 0, n = 0;
  uint64_t bidx = 0, acc = 0, p = 0, end = 0, sp = 0;
  uint8_t* outp = a.raw_out;
  const uint8_t* dictp = a.raw_out;
  uint32_t rem = 0, cq = 0, D = 1, span = 0;                          // match / stored-block copy in progress
  uint32_t pn = 0; uint64_t pc0 = 0, pc1 = 0;                         // match piece loaded last trip, appended this trip
  uint32_t pback = 0;                                                 // fix-up of a load pulled back from a buffer's end
  uint64_t w0 = 0, w1 = 0; uint32_t wo = 0;                           // 16 stream bytes; byte p of the stream at offset wo
  uint64_t ob0 = 0, ob1 = 0; uint32_t oc = 0, opos = 0;   // output bytes [opos, opos + oc) not stored yet
  uint32_t litA = 0;                                                  // 0x100 | short-code literal taken in front of it (slot A)
  uint32_t lit = 0;                                                   // 0x100 | literal decoded this trip

#ifdef HMSE_DIAG
  uint64_t cyc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  for (uint32_t trip = 0; trip < 0x7FFFFFF0u; trip++) {
    LK_T0();
#ifdef HMSE_DIAG
    cyc[0]++;
#endif
    // ---- 1. publish finished streams -----------------------------------------------------------------------------
    if (__ballot(st == ST_FIN)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      if (st == ST_FIN) {
        __hip_atomic_store(&a.done[k], bad ? 2u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a.ok) a.ok[k] = bad ? 0 : 1;
        if (bad) atomicOr(a.status, 1u);
        st = ST_WAIT; need_pull = true;
      }
    }
    LK_LAP(2);
    // ---- 2. header round ---------------------------------------------------------------------------------------------
    const uint32_t n_wait = (uint32_t)__builtin_popcountll(__ballot(st == ST_WAIT));
    const uint64_t m_run = __ballot(st == ST_DEC && !blocked);
    if (n_wait >= HDR_MIN || (n_wait != 0 && m_run == 0)) {
#ifdef HMSE_DIAG
      cyc[1]++;
      uint64_t ts_ = __builtin_readcyclecounter();
#endif
      if (st == ST_WAIT) {
        if (need_pull) {
          // two passes over the records inside one launch: first everything that needs no dictionary, then the DELTA
          // records — whose bases are then decoded already (or, for a chain of DELTAs and at the seam of the passes, were
          // pulled earlier and are being decoded).  In plain index order a DELTA record sits right behind its base (near-
          // duplicates are neighbours) and its lane idled for the whole of the base's decode: 22 of 64 lanes on average.
          for (;;) {
            if (!pass2) {
              k = atomicAdd(a.counter, 1u);
              if (k >= a.n_sel) { pass2 = true; continue; }
              if (a.kind[k] == HMSE_KIND_DELTA) continue;
            } else {
              k = atomicAdd(a.counter + 1, 1u);
              if (k >= a.n_sel) { st = ST_DONE; break; }
              if (a.kind[k] != HMSE_KIND_DELTA) continue;
            }
            break;
          }
          if (st != ST_DONE) {
            need_pull = false;
            const uint64_t o0 = a.raw_off[k], o1 = a.raw_off[k + 1];
            L = (uint32_t)(o1 - o0);
            outp = a.raw_out + o0;
            bad = o1 < o0 || o1 > a.raw_cap || o1 - o0 > 0x7FFFFFFFull;
            blocked = false; dictp = a.raw_out; Dl = 0; polls = 0;
            if (!bad && a.kind[k] == HMSE_KIND_DELTA) {
              const int64_t b = a.base ? a.base[k] : -1;
              if (b < 0 || (uint64_t)b >= k) bad = true;
              else { blocked = true; bidx = (uint64_t)b; }
            }
            const uint64_t s0 = a.stream_off[k];
            const uint64_t s1 = a.stream_len ? s0 + a.stream_len[k] : a.stream_off[k + 1];
            if (s1 < s0 || s1 > a.streams_bytes) bad = true;
            p = s0; end = bad ? s0 : s1; acc = 0; n = 0; pos = 0; last = false; eob = false;
            rem = 0; pn = 0; oc = 0; opos = 0; force = 0;
            const uint64_t bits = 8ull * (end - s0) + 64;
            budget = bits > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)bits;
          }
        }
        LK_SUB(12);
        if (st == ST_WAIT && bad) st = ST_FIN;
        if (st == ST_WAIT) {
          // -- block header (RFC 1951 §3.2.3) --
#define HREFILL() do { const uint64_t w_ = load8_at(a.streams, p, a.streams_bytes); acc |= w_ << n; const uint32_t adv_ = (63u - n) >> 3; p += adv_; n += adv_ * 8u; } while (0)
#define HTAKE(kk, dst) do { if (n < (kk)) HREFILL(); dst = (uint32_t)acc & ((1u << (kk)) - 1u); acc >>= (kk); n -= (kk); } while (0)
          if (budget-- == 0) bad = true;
          uint32_t hdr; HTAKE(3, hdr);
          last = (hdr & 1u) != 0;
          const uint32_t type = hdr >> 1;
          eob = false;
          if (bad) {
          } else if (type == 3) bad = true;
          else if (type == 0) {  // stored
            const uint32_t sk = n & 7u; acc >>= sk; n -= sk;
            uint32_t len, nlen; HTAKE(16, len); HTAKE(16, nlen);
            const uint64_t src = p - (n >> 3);
            if ((len ^ nlen) != 0xFFFFu || pos + len > L || src + len > end) bad = true;
            else {
              stored = true; sp = src; rem = len; cq = pos; pos += len;
              p = src + len; acc = 0; n = 0; eob = true;
            }
          } else {
            uint32_t nlit = 288, ndist = 32;
            const bool fixed = type == 1;
            uint32_t PC[8];
#pragma unroll
            for (int j = 0; j < 8; j++) PC[j] = 0;
            uint64_t cs0 = 0, cs1 = 0;             // code-length symbols sorted by (length, symbol): 5 bits each, 12 per word
            // A dynamic header is read from LDS: the next 288 stream bytes are staged in the (not yet needed) symbol area
            // with 18 loads in flight at once — read from global memory it costs a full round trip every few symbols,
            // ~300 of them in a row.  B0 = stream byte of LDS byte 0; lp = next LDS byte to enter the window.
            bool staged = false; uint64_t B0 = 0; uint32_t lp = 0;
#define LREFILL() do {                                                                                                   \
              if (staged && lp + 8 <= 288) {                                                                               \
                const uint32_t k_ = lp >> 2, r_ = (lp & 3u) * 8u;                                                          \
                const uint64_t lo_ = (uint64_t)my[k_ * 64] | ((uint64_t)my[(k_ + 1) * 64] << 32);                          \
                const uint64_t hi_ = my[(k_ + 2 < 72 ? k_ + 2 : 71) * 64];                                                 \
                const uint64_t w_ = r_ ? (lo_ >> r_) | (hi_ << (64u - r_)) : lo_;                                          \
                acc |= w_ << n; const uint32_t adv_ = (63u - n) >> 3; lp += adv_; n += adv_ * 8u;                          \
              } else {                                                                                                     \
                if (staged) { p = B0 + lp; staged = false; }                                                               \
                HREFILL();                                                                                                 \
              }                                                                                                            \
            } while (0)
            if (!fixed) {
              uint32_t t5; HTAKE(5, t5); nlit = t5 + 257; HTAKE(5, t5); ndist = t5 + 1;
              uint32_t ncl; HTAKE(4, ncl); ncl += 4;
              if (nlit > 286 || ndist > 30) bad = true;
              {
                const uint64_t bitpos = 8ull * p - n;
                B0 = bitpos >> 3;
                if (!bad && B0 + 288 <= a.streams_bytes) {
                  u32x4 sv[18];
#pragma unroll
                  for (int j = 0; j < 18; j++) __builtin_memcpy(&sv[j], a.streams + B0 + 16 * j, 16);
#pragma unroll
                  for (int j = 0; j < 18; j++) { my[(4 * j) * 64] = sv[j].x; my[(4 * j + 1) * 64] = sv[j].y; my[(4 * j + 2) * 64] = sv[j].z; my[(4 * j + 3) * 64] = sv[j].w; }
                  staged = true; lp = 0; acc = 0; n = 0;
                  LREFILL();
                  const uint32_t skip = (uint32_t)bitpos & 7u; acc >>= skip; n -= skip;
                }
              }
              uint64_t cl57 = 0;   // code-length code lengths, 3 bits per symbol
              for (uint32_t i = 0; i < ncl; i++) {
                if (n < 3) LREFILL();
                const uint32_t v = (uint32_t)acc & 7u; acc >>= 3; n -= 3;
                // order of the code-length code lengths (RFC 1951 §3.2.7): 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
                const uint32_t o = i < 3 ? 16 + i : i == 3 ? 0 : (i & 1) ? (19 - i) >> 1 : 6 + (i >> 1);
                cl57 |= (uint64_t)v << (3 * o);
              }
              uint64_t cnt8 = 0;   // symbols per length, 8-bit fields
              for (uint32_t s = 0; s < 19; s++) { const uint32_t l = (uint32_t)(cl57 >> (3 * s)) & 7u; cnt8 += 1ull << (8 * l); }
              // limits and offsets per length; the code-length code must be complete (or empty), as stock zlib demands
              uint32_t lim = 0, off = 0, maxl = 0; int32_t left = 1; bool over = false;
              uint64_t next8 = 0;
              PC[0] = 0u | (1u << 8);
#pragma unroll
              for (int l = 1; l <= 7; l++) {
                const uint32_t c = (uint32_t)(cnt8 >> (8 * l)) & 0xFFu;
                next8 |= (uint64_t)off << (8 * l);
                lim += c << (15 - l); off += c;
                left = (left << 1) - (int32_t)c;
                if (left < 0) over = true;
                if (c) maxl = l;
                PC[l] = (lim << 16) | ((uint32_t)(l + 1) << 8) | off;
              }
              if (over || (maxl != 0 && left > 0)) bad = true;
              if (!bad)
                for (uint32_t s = 0; s < 19; s++) {
                  const uint32_t l = (uint32_t)(cl57 >> (3 * s)) & 7u;
                  if (l) {
                    const uint32_t o = (uint32_t)(next8 >> (8 * l)) & 0xFFu; next8 += 1ull << (8 * l);
                    if (o < 12) cs0 |= (uint64_t)s << (5 * o); else cs1 |= (uint64_t)s << (5 * (o - 12));
                  }
                }
            }
            LK_SUB(13);
            // -- code lengths of the two alphabets.  They are needed twice, in order (counted per length now, placed by
            // rank afterwards): eight 4-bit lengths per word go through a 40-register FIFO — registers cannot be indexed
            // by a lane, but a queue only ever moves by one.  Counts per length live in packed registers as well:
            // literal/length and literals-only counts in 16-bit fields (four lengths per 64-bit word), distance counts in
            // 8-bit fields.  LDS holds nothing but the sorted literal/length symbols: 288 bytes per lane.
            uint32_t LF[40];
#pragma unroll
            for (int j = 0; j < 40; j++) LF[j] = 0;
#define LF_PUSH(wd) do { _Pragma("unroll") for (int j_ = 39; j_ > 0; j_--) LF[j_] = LF[j_ - 1]; LF[0] = (wd); } while (0)
            uint64_t lc[4] = {0, 0, 0, 0}, nl[4] = {0, 0, 0, 0}, dc[2] = {0, 0};
            const uint32_t tot = nlit + ndist;
            uint32_t i = 0, prev = 0, nw = 0, npush = 0;
            while (i < tot && !bad) {
              uint32_t rep = 1, val;
              if (fixed) val = i < 144 ? 8u : i < 256 ? 9u : i < 280 ? 7u : i < 288 ? 8u : 5u;
              else {
                if (n < 24) LREFILL();
                const uint32_t X = __builtin_bitreverse32((uint32_t)acc & 0x7FFFu) >> 17;
                uint32_t sel; IFL_SCAN(X, PC, 7, sel);
                const uint32_t cl = (sel >> 8) & 0xFFu;
                if (cl > 7) { bad = true; break; }
                const uint32_t ci = ((sel & 0xFFu) + ((X - (sel >> 16)) >> (15 - cl))) & 31u;
                const uint32_t s = (uint32_t)(ci < 12 ? cs0 >> (5 * ci) : cs1 >> (5 * (ci < 24 ? ci - 12 : 0))) & 31u;
                acc >>= cl; n -= cl;
                val = s;
                if (s == 16) { if (i == 0) { bad = true; break; } rep = 3 + ((uint32_t)acc & 3u); acc >>= 2; n -= 2; val = prev; }
                else if (s == 17) { rep = 3 + ((uint32_t)acc & 7u); acc >>= 3; n -= 3; val = 0; }
                else if (s == 18) { rep = 11 + ((uint32_t)acc & 127u); acc >>= 7; n -= 7; val = 0; }
                if (s > 18 || budget-- == 0) { bad = true; break; }
              }
              if (i + rep > tot) { bad = true; break; }
              // the run's counts in one step (a per-symbol loop would run to the longest run among the lanes of the round:
              // a 138-symbol run of zeros in one lane made all of them spin), then its nibbles a word at a time
              if (val) {
                const uint32_t n_lit = i >= nlit ? 0u : (nlit - i < rep ? nlit - i : rep);
                const uint32_t n_lo = i >= 256 ? 0u : (256 - i < rep ? 256 - i : rep);
                const uint32_t sh16 = 16 * (val & 3u);
                const uint64_t inc_lit = (uint64_t)n_lit << sh16, inc_lo = (uint64_t)n_lo << sh16;
                const uint32_t r4 = val >> 2;
#pragma unroll
                for (int q = 0; q < 4; q++) { lc[q] += r4 == (uint32_t)q ? inc_lit : 0ull; nl[q] += r4 == (uint32_t)q ? inc_lo : 0ull; }
                const uint64_t incd = (uint64_t)(rep - n_lit) << (8 * (val & 7u));
                dc[0] += val < 8 ? incd : 0ull; dc[1] += val >= 8 ? incd : 0ull;
              }
              while (rep) {
                const uint32_t o = i & 7u, k = rep < 8 - o ? rep : 8 - o;
                const uint32_t fill = k == 8 ? 0xFFFFFFFFu : ((1u << (4 * k)) - 1u) << (4 * o);
                nw |= (val * 0x11111111u) & fill;
                i += k; rep -= k;
                if ((i & 7u) == 0) { LF_PUSH(nw); nw = 0; npush++; }
              }
              prev = val;
            }
            if (!bad) {
              if (i & 7u) { LF_PUSH(nw); npush++; }
              for (; npush < 40; npush++) LF_PUSH(0u);                  // word 0 of the lengths now sits in LF[39]
              // lens[256] == 0: no end-of-block code.  Word 32 is 7 pushes above the last one.
              if ((LF[7] & 0xFu) == 0) bad = true;
            }
            if (staged) { p = B0 + lp; staged = false; }                 // back to the stream itself
            LK_SUB(14);
            uint64_t nxl[4] = {0, 0, 0, 0}, nxd[2] = {0, 0};             // running offset per length (the counts' layout)
            if (!bad) {
              // -- literal/length decode words --
              uint32_t lim = 0, off = 0, maxl = 0; int32_t left = 1; bool over = false;
              P[0] = 0; Q[0] = (1u << 9) | ((uint32_t)(nl[0] >> 16) & 0xFFFFu);
#pragma unroll
              for (int l = 1; l <= 15; l++) {
                const uint32_t c = (uint32_t)(lc[l >> 2] >> (16 * (l & 3))) & 0xFFFFu;
                nxl[l >> 2] |= (uint64_t)off << (16 * (l & 3));
                lim += c << (15 - l); off += c;
                left = (left << 1) - (int32_t)c;
                if (left < 0) over = true;
                if (c) maxl = l;
                const uint32_t nlo_next = l < 15 ? (uint32_t)(nl[(l + 1) >> 2] >> (16 * ((l + 1) & 3))) & 0xFFFFu : 0u;
                P[l] = (lim << 16) | (l < 15 ? off : 0u);
                Q[l] = ((uint32_t)(l + 1) << 9) | (l < 15 ? off + nlo_next : 0u);
              }
              if (over || (left > 0 && maxl != 1)) bad = true;
              // -- distance decode words --
              lim = 0; off = 0; maxl = 0; left = 1; over = false;
              PD[0] = 1u << 8;
#pragma unroll
              for (int l = 1; l <= 15; l++) {
                const uint32_t c = (uint32_t)(dc[l >> 3] >> (8 * (l & 7))) & 0xFFu;
                nxd[l >> 3] |= (uint64_t)off << (8 * (l & 7));
                lim += c << (15 - l); off += c;
                left = (left << 1) - (int32_t)c;
                if (left < 0) over = true;
                if (c) maxl = l;
                PD[l] = (lim << 16) | ((uint32_t)(l + 1) << 8) | (l < 15 ? off : 0u);
              }
              if (over || (maxl != 0 && left > 0 && maxl != 1)) bad = true;
            }
            if (!bad) {
              // -- symbols sorted by (length, symbol): literal/length symbols into LDS, distance symbols into registers --
              ds0 = 0; ds1 = 0; ds2 = 0;
              const uint32_t nwords = (tot + 7) >> 3;
              for (uint32_t w = 0; w < nwords; w++) {
                const uint32_t word = LF[39];
                LF_PUSH(0u);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                  const uint32_t s = w * 8 + j;
                  const uint32_t v = (word >> (4 * j)) & 0xFu;
                  if (!v) continue;                                       // (symbols >= tot have length 0)
                  if (s < nlit) {
                    const uint32_t r4 = v >> 2, sh = 16 * (v & 3u);
                    const uint64_t cur = r4 == 0 ? nxl[0] : r4 == 1 ? nxl[1] : r4 == 2 ? nxl[2] : nxl[3];
                    const uint32_t o = (uint32_t)(cur >> sh) & 0xFFFFu;
                    const uint64_t inc = 1ull << sh;
#pragma unroll
                    for (int q = 0; q < 4; q++) nxl[q] += r4 == (uint32_t)q ? inc : 0ull;
                    LB(W_LSYM * 4 + (o < 288 ? o : 0)) = (uint8_t)s;
                  } else {
                    const uint32_t sh = 8 * (v & 7u);
                    const uint32_t o = (uint32_t)((v < 8 ? nxd[0] : nxd[1]) >> sh) & 0xFFu;
                    const uint64_t inc = 1ull << sh;
                    nxd[0] += v < 8 ? inc : 0ull; nxd[1] += v >= 8 ? inc : 0ull;
                    const uint64_t dsym = (uint64_t)(s - nlit) << (5 * (o % 12u));
                    ds0 |= o < 12 ? dsym : 0ull; ds1 |= (o >= 12 && o < 24) ? dsym : 0ull; ds2 |= (o >= 24 && o < 36) ? dsym : 0ull;
                  }
                }
              }
              HREFILL();
              w0 = load8_at(a.streams, p, a.streams_bytes); w1 = load8_at(a.streams, p + 8, a.streams_bytes); wo = 0;
            }
#undef LF_PUSH
#undef LREFILL
          }
          LK_SUB(15);
          if (bad) { st = ST_FIN; rem = 0; pn = 0; oc = 0; }
          else st = ST_DEC;
        }
      }
    }
    LK_LAP(3);
    if (__ballot(st != ST_DONE) == 0) break;
    // ---- 3. streams waiting for their base chunk -------------------------------------------------------------------
    if ((trip & 7u) == 0 && __ballot(st == ST_DEC && blocked)) {   // a poll waits for a load: not in every trip
      bool woke = false;
      if (st == ST_DEC && blocked) {
        const uint32_t f = __hip_atomic_load(&a.done[bidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (f == 1u) {
          const uint64_t b0 = a.raw_off[bidx], b1 = a.raw_off[bidx + 1];
          uint64_t dl = b1 - b0;
          dictp = a.raw_out + b0;
          if (dl > 32768) { dictp += dl - 32768; dl = 32768; }
          Dl = (uint32_t)dl; blocked = false; woke = true;
        } else if (f == 2u) { bad = true; }
        else if (++polls >= POLL_MAX) { bad = true; atomicOr(a.status, 2u); }
        if (bad) { blocked = false; st = ST_FIN; rem = 0; pn = 0; oc = 0; }
      }
      if (__ballot(woke)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      else if (m_run == 0 && n_wait == 0) __builtin_amdgcn_s_sleep(32);
    }
    LK_LAP(4);
    // ---- 4. one token ----------------------------------------------------------------------------------------------
#ifdef HMSE_DIAG
    cyc[7] += (uint64_t)__builtin_popcountll(__ballot(st == ST_DEC && !blocked && rem == 0 && !eob));
    cyc[8] += (uint64_t)__builtin_popcountll(__ballot(st == ST_DEC && !blocked && rem != 0));   // copying a long match
    cyc[9] += (uint64_t)__builtin_popcountll(__ballot(st == ST_WAIT));                           // waiting for a header round
    cyc[10] += (uint64_t)__builtin_popcountll(__ballot(st == ST_DONE));                          // out of work
    cyc[11] += (uint64_t)__builtin_popcountll(__ballot(st == ST_DEC && blocked));                // waiting for a base chunk
#endif
    const bool dec = st == ST_DEC && !blocked && rem == 0;
    if (dec && eob) {                                                   // end of a block: rare, a real branch
      if (pn == 0) {
        if (!last) st = ST_WAIT;                                        // the output buffer carries over into the next block
        else if (oc != 0) force = 1;                                    // flushed in this trip's step 5
        else { if (pos != L || p - (n >> 3) != end) bad = 1; st = ST_FIN; }
      }
    }
    if (dec && !eob) {
      // Straight-line code: every lane computes the literal, the end-of-block and the length/distance outcome and keeps
      // the one its symbol selects (with 64 independent streams every path is taken in every trip anyway; as nested ifs
      // the step was half scalar mask bookkeeping and branches).  A lane whose symbol is no length consumes 0 extra bits
      // and a 0-bit distance code.
      // Slot A: two thirds of all tokens are literals, and the frequent ones have short codes.  A literal whose code has
      // at most 9 bits is taken here on top (9 of the 15 compare/select pairs), provided 48 bits remain for the full
      // token step below, which then decodes the token AFTER it: ~1.5 tokens per trip instead of 1.
      {
        const uint32_t XA = __builtin_bitreverse32((uint32_t)acc & 0x7FFFu) >> 17;
        const uint32_t xk = (XA << 16) | 0xFFFFu;
        uint32_t sa = P[0], sq = Q[0];
#pragma unroll
        for (int j = 1; j <= 8; j++) { const bool ge = xk >= P[j]; sa = ge ? P[j] : sa; sq = ge ? Q[j] : sq; }
        const uint32_t la = sq >> 9;                                     // 1..9, the code's length if xk < P[9]
        const uint32_t ia = ((sa & 0xFFFFu) + ((XA - (sa >> 16)) >> (15 - la))) & 0x1FFu;
        const uint32_t ba = LB(W_LSYM * 4 + (ia < 288 ? ia : 0));
        const bool take = xk < P[9] && ia < (sq & 0x1FFu) && la + 48 <= n && pos < L && budget > 1;
        litA = take ? ba | 0x100u : 0u;
        const uint32_t da = take ? la : 0u;
        acc >>= da; n -= da;
        pos += take ? 1u : 0u; budget -= take ? 1u : 0u;
      }
      uint32_t err = budget == 0 ? 1u : 0u;
      budget -= 1;
      const uint32_t X = __builtin_bitreverse32((uint32_t)acc & 0x7FFFu) >> 17;   // the window was refilled in step 5
      uint32_t sel, selq;
      {
        const uint32_t xk = (X << 16) | 0xFFFFu;
        sel = P[0]; selq = Q[0];
#pragma unroll
        for (int j = 1; j <= 15; j++) { const bool ge = xk >= P[j]; sel = ge ? P[j] : sel; selq = ge ? Q[j] : selq; }
      }
      const uint32_t l1 = selq >> 9;
      err |= l1 > 15 ? 1u : 0u;
      const uint32_t l1c = l1 > 15 ? 15u : l1;
      const uint32_t idx = ((sel & 0xFFFFu) + ((X - (sel >> 16)) >> (15 - l1c))) & 0x1FFu;
      const uint32_t sym = (uint32_t)LB(W_LSYM * 4 + (idx < 288 ? idx : 0)) | (idx >= (selq & 0x1FFu) ? 256u : 0u);
      acc >>= l1c; n -= l1c;
      const bool is_lit = sym < 256, is_len = sym > 256;
      err |= sym > 285 ? 1u : 0u;
      // length and distance (RFC 1951 §3.2.5)
      const uint32_t lcx = is_len ? (sym - 257u) & 31u : 0u;
      const bool lshort = lcx < 8 || lcx >= 28;
      const uint32_t e1 = lshort ? 0u : (lcx - 4) >> 2;
      const uint32_t lb = lcx < 8 ? 3 + lcx : lcx >= 28 ? 258u : 3 + ((4 + (lcx & 3)) << e1);
      const uint32_t len = lb + ((uint32_t)acc & ((1u << e1) - 1u));
      acc >>= e1; n -= e1;
      const uint32_t X2 = __builtin_bitreverse32((uint32_t)acc & 0x7FFFu) >> 17;
      uint32_t seld; IFL_SCAN(X2, PD, 15, seld);
      const uint32_t l2r = (seld >> 8) & 0xFFu;
      const uint32_t l2c = l2r > 15 ? 15u : l2r;
      const uint32_t di = ((seld & 0xFFu) + ((X2 - (seld >> 16)) >> (15 - l2c))) & 31u;
      const uint32_t dj = di < 12 ? di : di < 24 ? di - 12 : di - 24;
      const uint32_t ds = (uint32_t)((di < 12 ? ds0 : di < 24 ? ds1 : ds2) >> (5 * dj)) & 31u;
      err |= (is_len && (l2r > 15 || ds > 29)) ? 1u : 0u;
      const uint32_t l2 = is_len ? l2c : 0u;
      acc >>= l2; n -= l2;
      const uint32_t e2 = (is_len && ds >= 4) ? ((ds >> 1) - 1u) & 15u : 0u;
      const uint32_t dist = ds < 4 ? 1 + ds : 1 + ((2 + (ds & 1)) << e2) + ((uint32_t)acc & ((1u << e2) - 1u));
      acc >>= e2; n -= e2;
      err |= (is_lit && pos >= L) ? 1u : 0u;
      err |= (is_len && (pos + len > L || dist > pos + Dl)) ? 1u : 0u;
      if (!err) {
        lit = is_lit ? sym | 0x100u : 0u;                                // appended to the output buffer in step 5
        eob = sym == 256 ? 1u : 0u;
        rem = is_len ? len : 0u;
        cq = is_len ? pos : cq; D = is_len ? dist : D; span = is_len ? dist : span; stored = is_len ? 0u : stored;
        pos += is_lit ? 1u : is_len ? len : 0u;
      } else { bad = 1; st = ST_FIN; rem = 0; pn = 0; oc = 0; litA = 0; }
    }
    LK_LAP(5);
    // ---- 5. every memory operation of the trip, in one cluster: what it waits for was issued a whole trip ago ----------
    // What this kernel pays for is the NUMBER of lane-requests: every lane addresses its own cache line, and such an
    // instruction costs the CU ~8 cycles per active lane whatever its width (tools/ubench/gmem_divergent.hip: byte store
    // 517, unaligned dwordx4 store 516, dwordx4 load 330..470 cycles per instruction).  So:
    //   * output goes through a 16-byte register buffer per lane (literals and match pieces appended in stream order) and
    //     leaves as ONE 16-byte store when the next piece does not fit — the bytes behind the valid ones are garbage that
    //     lands on positions this lane writes later anyway (never past the chunk's end: there the store is exact);
    //   * the stream is read 16 bytes at a time into a register window, reloaded only when fewer than 8 bytes are left in it;
    //   * a match piece is one 16-byte load (15 bytes used), issued a trip before it is appended.
    // gfx9 counts loads and stores in ONE counter, so a wait for a load is a wait for every store before it too: all
    // consumers of last trip's loads sit at the top of the cluster (the empty asm gives the compiler one unconditional
    // place for its wait; left alone it waits again in front of every conditional use, i.e. behind this trip's stores),
    // then come the stores, then the new loads.  A load whose address would run over the end of its buffer is pulled back
    // and its value shifted when used (a branch around a load merges a loaded with a computed value, and the compiler
    // waits for the load on the spot).
    asm volatile("" : "+v"(pc0), "+v"(pc1), "+v"(w0), "+v"(w1));
    const bool fill = st == ST_DEC && !blocked && !eob;
    {
      const uint32_t sh = wo * 8u;                                     // stream byte p sits at byte wo of the window
      const uint64_t x = sh < 64 ? w0 : w1, y = sh < 64 ? w1 : 0ull;
      const uint32_t s6 = sh & 63u;
      uint64_t val = (x >> s6) | (s6 ? y << (64u - s6) : 0ull);
      val = sh >= 128 ? 0ull : val;
      const uint32_t adv = fill ? (63u - n) >> 3 : 0u;
      acc |= fill ? val << (n & 63u) : 0ull;      // (n <= 63 wherever `fill` holds; the mask keeps the shift defined elsewhere)
      p += adv; n += adv * 8u; wo += adv;
    }
#ifdef HMSE_DIAG
    if (a.dflags & 1u) { lit = 0; litA = 0; }
    if (a.dflags & 2u) pn = 0;
#endif
    if (pn && pback) {   // the load was pulled back from the end of the buffer: drop the bytes in front
      const uint32_t t = pback * 8;
      if (t >= 64) { pc0 = pc1 >> (t - 64); pc1 = 0; }
      else { pc0 = (pc0 >> t) | (pc1 << (64 - t)); pc1 >>= t; }
      pback = 0;
    }
    const uint32_t m_in = pn + (litA ? 1u : 0u) + (lit ? 1u : 0u);
    if (oc != 0 && (force || oc + m_in > 16)) {
      uint8_t* const d = outp + opos;
      if (opos + 16 <= L) { __builtin_memcpy(d, &ob0, 8); __builtin_memcpy(d + 8, &ob1, 8); }
      else if (oc >= 8) {                                              // the chunk's last bytes: exactly oc of them
        __builtin_memcpy(d, &ob0, 8);
        if (oc > 8) {
          const uint32_t t = (oc - 8) * 8;
          const uint64_t v = t == 64 ? ob1 : (ob0 >> t) | (ob1 << (64 - t));
          __builtin_memcpy(d + oc - 8, &v, 8);
        }
      } else if (oc >= 4) {
        const uint32_t v0 = (uint32_t)ob0;
        __builtin_memcpy(d, &v0, 4);
        if (oc > 4) { const uint32_t v1 = (uint32_t)(ob0 >> (8 * (oc - 4))); __builtin_memcpy(d + oc - 4, &v1, 4); }
      } else {
        d[0] = (uint8_t)ob0;
        if (oc >= 2) d[1] = (uint8_t)(ob0 >> 8);
        if (oc == 3) d[2] = (uint8_t)(ob0 >> 16);
      }
      opos += oc; oc = 0; force = 0;
    }
    // append: ob = (ob & bytes below oc) | x << 8*oc; whatever x carries above its valid bytes lands above the new oc —
    // so appending 0 bytes of anything is harmless and both appends run unconditionally
#define OB_APPEND(x0, x1, cnt) do {                                                                          \
      const uint32_t t_ = oc * 8u, u_ = t_ & 63u;                                                            \
      const uint64_t keep_ = (1ull << u_) - 1ull, up_ = (x0) << u_;                                          \
      const uint64_t carry_ = u_ ? (x0) >> (64u - u_) : 0ull;                                                \
      const uint64_t n0_ = (ob0 & keep_) | up_, n1a_ = ((x1) << u_) | carry_, n1b_ = (ob1 & keep_) | up_;    \
      ob1 = t_ < 64 ? n1a_ : t_ < 128 ? n1b_ : ob1;                                                          \
      ob0 = t_ < 64 ? n0_ : ob0;                                                                             \
      oc += (cnt);                                                                                           \
    } while (0)
    OB_APPEND(pc0, pc1, pn); pn = 0;
    {   // the trip's literals (slot A's, then the token step's) as one append
      const uint64_t lb = litA ? (uint64_t)((litA & 0xFFu) | ((lit & 0xFFu) << 8)) : (uint64_t)(lit & 0xFFu);
      OB_APPEND(lb, 0ull, ((litA ? 1u : 0u) + (lit ? 1u : 0u)));
      litA = 0; lit = 0;
    }
#undef OB_APPEND
    if (fill && wo > 8) {                                               // fewer than 8 bytes left in the window
      const uint64_t q = p + 16 <= a.streams_bytes ? p : a.streams_bytes - 16;   // the host side guarantees >= 16 bytes
      const uint64_t back = p - q;
      wo = back > 31 ? 31u : (uint32_t)back;
      __builtin_memcpy(&w0, a.streams + q, 8); __builtin_memcpy(&w1, a.streams + q + 8, 8);
    }
    if (rem) {
      uint32_t nb = rem < 14 ? rem : 14;                                 // + two literals of the next trip <= 16
      const uint8_t* s; const uint8_t* lim;
      bool clash = false;
      if (stored) { s = a.streams + sp; lim = a.streams + a.streams_bytes; }
      else {
        nb = nb < D ? nb : D;
        const int32_t sv = (int32_t)cq - (int32_t)D;                  // < 0: inside the dictionary
        if (sv < 0) { const uint32_t room = (uint32_t)(-sv); nb = nb < room ? nb : room; s = dictp + (int64_t)Dl + sv; }
        else { s = outp + sv; clash = oc != 0 && (uint32_t)sv + nb > opos; }   // source bytes still in the register buffer
        lim = a.raw_out + a.raw_cap;
      }
      if (clash) force = 1;                                             // flushed next trip, loaded the trip after
      else {
        pback = s + 16 <= lim ? 0u : (uint32_t)(s + 16 - lim);         // <= 15: s + nb <= lim (buffers hold >= 16 bytes)
#ifdef HMSE_DIAG
        if (!(a.dflags & 4u))
#endif
        { __builtin_memcpy(&pc0, s - pback, 8); __builtin_memcpy(&pc1, s - pback + 8, 8); }
        pn = nb; cq += nb; rem -= nb;
        if (stored) sp += nb;
        else { span += nb; if (D < 16 && 2 * D <= span) D *= 2; }
      }
    }
    LK_LAP(6);
  }
#ifdef HMSE_DIAG
  if (a.trace && lane == 0)
    for (int i = 0; i < 16; i++) atomicAdd((unsigned long long*)a.trace + i, (unsigned long long)cyc[i]);
#endif
#undef LB
#undef HREFILL
#undef HTAKE
}

}  // namespace ifl

namespace ifl {

// chunk i of the original data = raw bytes of the stored chunk it points to (README.md:1635-1669)
__global__ __launch_bounds__(256) void assemble_kernel(const uint64_t* __restrict__ cuts, uint64_t n_chunks,
                                                        const uint64_t* __restrict__ slot_of_chunk, uint64_t n_slots,
                                                        const uint64_t* __restrict__ raw_off, const uint8_t* __restrict__ raw,
                                                        uint8_t* __restrict__ data_out, uint64_t n, uint32_t* status) {
  const uint64_t i = blockIdx.x;
  if (i >= n_chunks) return;
  const uint64_t s = slot_of_chunk[i];
  const uint64_t d0 = cuts[i], d1 = cuts[i + 1];
  if (s >= n_slots || d1 < d0 || d1 > n || raw_off[s + 1] - raw_off[s] != d1 - d0) {  // inconsistent manifest
    if (threadIdx.x == 0) atomicOr(status, 1u);
    return;
  }
  const uint64_t len = d1 - d0;
  const uint8_t* src = raw + raw_off[s];
  uint8_t* dst = data_out + d0;
  for (uint64_t b = (uint64_t)threadIdx.x * 16; b < len; b += 256 * 16) {
    if (b + 16 <= len) { const uint4 v = load_u4_unaligned(src + b); __builtin_memcpy(dst + b, &v, 16); }
    else for (uint64_t j = b; j < len; j++) dst[j] = src[j];
  }
}

}  // namespace ifl

#ifdef HMSE_DIAG
static uint32_t* g_ifl_trace = nullptr;
// diagnostics hook, not part of the ABI: progress words in host-visible memory (tools/inflate_debug.py)
extern "C" void hmsedbg_inflate_trace(void* p) { g_ifl_trace = (uint32_t*)p; }
#else
static uint32_t* const g_ifl_trace = nullptr;
#endif

// 0 = choose by stream count, 1 = one stream per wavefront, 2 = one stream per lane
static std::atomic<int> g_ifl_mode{0};   // process-wide knob (include/hmse.h); atomic: a host thread may flip it while another decodes
constexpr uint64_t HMSE_INFLATE_WIDE_MIN = 49152;
extern "C" int hmse_l1_inflate_mode(int mode) {
  if (mode < 0 || mode > 2) return HMSE_EINVAL;
  g_ifl_mode = mode;
  return HMSE_OK;
}

size_t hmse_l1_inflate_workspace_bytes_impl(uint64_t n_sel) { return 256 + hmse_align_up((size_t)n_sel * 4, 256); }

extern "C" int hmse_l1_inflate(const uint8_t* streams, uint64_t streams_bytes, const uint64_t* stream_off, const uint32_t* stream_len,
                               const uint8_t* kind, const int64_t* base, uint64_t n_sel, const uint64_t* raw_off, uint8_t* raw_out,
                               uint64_t raw_cap, uint8_t* ok, uint32_t* status, void* ws, size_t ws_bytes, void* stream_) {
  using namespace ifl;
  if (!status) return HMSE_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  (void)hipGetLastError();
  HMSE_FILL(status, 0, sizeof(uint32_t), stream);
  if (n_sel == 0) return HMSE_OK;
  if (!streams || !stream_off || !kind || !raw_off || !raw_out || n_sel > 0x7FFFFFFFull) return HMSE_EINVAL;
  if (!ws || ws_bytes < hmse_l1_inflate_workspace_bytes_impl(n_sel)) return HMSE_ENOSPC;
  HMSE_FILL(ws, 0, hmse_l1_inflate_workspace_bytes_impl(n_sel), stream);
  Args a;
  a.streams = streams; a.streams_bytes = streams_bytes; a.stream_off = stream_off; a.stream_len = stream_len;
  a.kind = kind; a.base = base; a.n_sel = (uint32_t)n_sel;
  a.raw_off = raw_off; a.raw_out = raw_out; a.raw_cap = raw_cap; a.status = status; a.ok = ok; a.trace = g_ifl_trace;
  a.dflags = 0;
#ifdef HMSE_DIAG
  if (const char* e = getenv("HMSE_IFL_DFLAGS")) a.dflags = (uint32_t)atoi(e);
#endif
  a.counter = (uint32_t*)ws; a.done = (uint32_t*)((uint8_t*)ws + 256);
  // few streams: one per wavefront (latency of a call = one stream's decode); many: one per lane (four times the
  // throughput once the chip's 65 536 lanes have a stream each, but a call takes as long as ~a lane's streams in sequence)
  const bool wide = (g_ifl_mode == 2 || (g_ifl_mode == 0 && n_sel >= HMSE_INFLATE_WIDE_MIN)) && streams_bytes >= 16 && raw_cap >= 16;
  PROF_BEGIN(HMSE_STAGE_L1_INFLATE, stream);
  if (wide) {
    uint64_t blocks = (n_sel + 63) / 64;
    if (blocks > 256 * 8) blocks = 256 * 8;  // persistent; 18 KiB of LDS each: eight per CU, two per SIMD
    l1_inflate_lanes_kernel<<<dim3((uint32_t)blocks), dim3(64), 0, stream>>>(a);
  } else {
    uint64_t blocks = (n_sel + NT / 64 - 1) / (NT / 64);
    if (blocks > 256 * 8) blocks = 256 * 8;  // persistent wavefronts; a waiting wavefront's base was pulled earlier, so it is running or done
    l1_inflate_kernel<<<dim3((uint32_t)blocks), dim3(NT), 0, stream>>>(a);
  }
  PROF_END(HMSE_STAGE_L1_INFLATE, stream);
  HMSE_LAUNCH_CHECK();
  return HMSE_OK;
}

extern "C" int hmse_read_assemble(const uint64_t* cuts, uint64_t n_chunks, const uint64_t* slot_of_chunk, uint64_t n_slots,
                                  const uint64_t* raw_off, const uint8_t* raw, uint8_t* data_out, uint64_t n, uint32_t* status,
                                  void* stream_) {
  if (!status) return HMSE_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  (void)hipGetLastError();
  HMSE_FILL(status, 0, sizeof(uint32_t), stream);
  if (n_chunks == 0) return HMSE_OK;
  if (!cuts || !slot_of_chunk || !raw_off || !raw || !data_out) return HMSE_EINVAL;
  if (n_chunks > 0x7FFFFFFFull) return HMSE_EINVAL;
  PROF_BEGIN(HMSE_STAGE_READ_ASSEMBLE, stream);
  ifl::assemble_kernel<<<dim3((uint32_t)n_chunks), dim3(256), 0, stream>>>(cuts, n_chunks, slot_of_chunk, n_slots, raw_off, raw, data_out, n, status);
  PROF_END(HMSE_STAGE_READ_ASSEMBLE, stream);
  HMSE_LAUNCH_CHECK();
  return HMSE_OK;
}
