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

size_t hmse_l1_inflate_workspace_bytes_impl(uint64_t n_sel) { return 256 + hmse_align_up((size_t)n_sel * 4, 256); }

extern "C" int hmse_l1_inflate(const uint8_t* streams, uint64_t streams_bytes, const uint64_t* stream_off, const uint32_t* stream_len,
                               const uint8_t* kind, const int64_t* base, uint64_t n_sel, const uint64_t* raw_off, uint8_t* raw_out,
                               uint64_t raw_cap, uint8_t* ok, uint32_t* status, void* ws, size_t ws_bytes, void* stream_) {
  using namespace ifl;
  if (!status) return HMSE_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  (void)hipGetLastError();
  HMSE_HIP(hipMemsetAsync(status, 0, sizeof(uint32_t), stream));
  if (n_sel == 0) return HMSE_OK;
  if (!streams || !stream_off || !kind || !raw_off || !raw_out || n_sel > 0x7FFFFFFFull) return HMSE_EINVAL;
  if (!ws || ws_bytes < hmse_l1_inflate_workspace_bytes_impl(n_sel)) return HMSE_ENOSPC;
  HMSE_HIP(hipMemsetAsync(ws, 0, hmse_l1_inflate_workspace_bytes_impl(n_sel), stream));
  Args a;
  a.streams = streams; a.streams_bytes = streams_bytes; a.stream_off = stream_off; a.stream_len = stream_len;
  a.kind = kind; a.base = base; a.n_sel = (uint32_t)n_sel;
  a.raw_off = raw_off; a.raw_out = raw_out; a.raw_cap = raw_cap; a.status = status; a.ok = ok; a.trace = g_ifl_trace;
  a.counter = (uint32_t*)ws; a.done = (uint32_t*)((uint8_t*)ws + 256);
  uint64_t blocks = (n_sel + NT / 64 - 1) / (NT / 64);
  if (blocks > 256 * 8) blocks = 256 * 8;  // persistent wavefronts; a waiting wavefront's base was pulled earlier, so it is running or done
  PROF_BEGIN(HMSE_STAGE_L1_INFLATE, stream);
  l1_inflate_kernel<<<dim3((uint32_t)blocks), dim3(NT), 0, stream>>>(a);
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
  HMSE_HIP(hipMemsetAsync(status, 0, sizeof(uint32_t), stream));
  if (n_chunks == 0) return HMSE_OK;
  if (!cuts || !slot_of_chunk || !raw_off || !raw || !data_out) return HMSE_EINVAL;
  if (n_chunks > 0x7FFFFFFFull) return HMSE_EINVAL;
  PROF_BEGIN(HMSE_STAGE_READ_ASSEMBLE, stream);
  ifl::assemble_kernel<<<dim3((uint32_t)n_chunks), dim3(256), 0, stream>>>(cuts, n_chunks, slot_of_chunk, n_slots, raw_off, raw, data_out, n, status);
  PROF_END(HMSE_STAGE_READ_ASSEMBLE, stream);
  HMSE_LAUNCH_CHECK();
  return HMSE_OK;
}
