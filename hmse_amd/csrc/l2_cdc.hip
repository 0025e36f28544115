// l2_cdc.hip — L2 content-defined chunking (Gear-hash FastCDC) for gfx950.
//
// Replaces rabin_slide() + the cut loop of the reference skeleton (README.md:2456-2464,
// 2475-2490).  Because the rolling state is never reset at a cut, "byte i
// is a cut candidate" is a pure function of the 64 bytes ending at i, so the work splits into
//   (1) l2_hash_kernel   — HBM-bound scan: one 128-byte strip per lane loaded with 16-byte loads
//                          (each cache line is consumed whole by its lane), 64-byte warm-up, Gear
//                          table in LDS, one v_min per byte into a running minimum (hits are
//                          located only in the rare 16-byte group whose minimum is below the
//                          threshold), candidate bitmaps in registers, wavefront prefix-scan
//                          compaction into a sorted list;
//   (2) l2_resolve_kernel — per segment, one wavefront walks the sorted candidates with
//                          ballots applying MIN / two-mask normalisation / MAX;
//   (3) l2_scan/l2_copy  — concatenate the per-segment cut lists.
#include "common.h"

// ---- compile-time Gear table (same generator as hmse_gear_table) ---------------------------
struct GearTable { uint64_t v[256]; };
static constexpr GearTable make_gear() {
  GearTable t{};
  uint64_t x = 0x484D53455F4C3247ull;  // "HMSE_L2G"
  for (int i = 0; i < 256; i++) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    t.v[i] = z ^ (z >> 31);
  }
  return t;
}
static constexpr GearTable kGearHost = make_gear();
__device__ const GearTable kGearDev = make_gear();

extern "C" void hmse_gear_table(uint64_t table[256]) { memcpy(table, kGearHost.v, sizeof kGearHost.v); }

// ---- geometry ---------------------------------------------------------------------------------
constexpr int L2_NT = 256;                     // threads per workgroup
#ifndef HMSE_L2_STRIP
#define HMSE_L2_STRIP 128
#endif
#ifndef HMSE_L2_COPIES
#define HMSE_L2_COPIES 4
#endif
// Two data paths were measured (tools/l2_sweep.py, 4 GiB): tile staged in LDS (define HMSE_L2_STAGED;
// 2.17 TB/s with one table copy, LDS-capacity-limited to 16 waves/CU) and direct per-lane strip loads
// (default; 2.29 TB/s, 28 waves/CU).
#ifndef HMSE_L2_STAGED
#define HMSE_L2_DIRECT 1
#endif
constexpr int L2_STRIP = HMSE_L2_STRIP;          // bytes per lane
constexpr int L2_TILE = L2_NT * L2_STRIP;      // 32 KiB of input per workgroup
[[maybe_unused]] constexpr int L2_PSTRIDE = L2_STRIP + 16;  // staged variant only: padded strip pitch, ds_read_b128 conflict-free
constexpr int L2_COPIES = HMSE_L2_COPIES;        // Gear table replicas: 1 measured best (occupancy beats bank spreading)
constexpr int L2_WORDS = L2_STRIP / 32;        // bitmap dwords per lane

struct TileInfo { unsigned long long base; uint32_t count; uint32_t pad; };

struct L2Header { unsigned long long cand_total; uint32_t status; uint32_t pad; };

template <int COPIES>
__device__ __forceinline__ uint64_t gear_lookup(const uint64_t* lg, uint32_t byte, uint32_t copy) {
  return lg[byte * COPIES + copy];
}

__global__ __launch_bounds__(L2_NT) void l2_hash_kernel(const uint8_t* __restrict__ data0, const uint64_t* __restrict__ data_off, uint64_t n,
                                                          uint32_t thr_l, uint32_t thr_s,
                                                          uint32_t* __restrict__ cand, uint64_t cand_cap,
                                                          TileInfo* __restrict__ tinfo, L2Header* hdr) {
  const uint8_t* __restrict__ data = data0 + (data_off ? *data_off : 0ull);   // (captured chain: the batch's offset lives in HBM)
#ifndef HMSE_L2_DIRECT
  __shared__ __attribute__((aligned(16))) uint8_t s_data[(L2_NT + 1) * L2_PSTRIDE];
#endif
  __shared__ uint64_t s_gear[256 * L2_COPIES];
  __shared__ uint32_t s_red[L2_NT / 64 + 1];
  __shared__ unsigned long long s_base;

  const uint32_t t = threadIdx.x;
  const uint64_t tile = blockIdx.x;
  const uint64_t t0 = tile * (uint64_t)L2_TILE;

  // Gear table replicas
#pragma unroll
  for (int c = 0; c < L2_COPIES; c++) s_gear[t * L2_COPIES + c] = kGearDev.v[t];

#ifdef HMSE_L2_DIRECT
  // Direct variant: every lane loads its own strip (plus the 64 bytes in front of it) straight into
  // registers with 16-byte loads.  A wave-instruction touches 64 different cache lines, but each line is
  // consumed completely by its lane's loads (L1 hits after the first), and without a staged tile only the
  // Gear table lives in LDS, so the CU can hold its full 32 waves.
  const uint64_t gs0 = t0 + (uint64_t)t * L2_STRIP;
  uint4 wm[4], r[L2_STRIP / 16];
  auto ld = [&](uint64_t g) -> uint4 {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (g + 16 <= n) v = load_u4_unaligned(data + g);
    else if (g < n) { uint8_t tmp[16]; for (int bb = 0; bb < 16; bb++) tmp[bb] = (g + bb < n) ? data[g + bb] : (uint8_t)0; __builtin_memcpy(&v, tmp, 16); }
    return v;
  };
#pragma unroll
  for (int j = 0; j < 4; j++) wm[j] = gs0 >= 64 ? ld(gs0 - 64 + 16 * j) : make_uint4(0, 0, 0, 0);
#pragma unroll
  for (int j = 0; j < L2_STRIP / 16; j++) r[j] = ld(gs0 + 16 * j);
  __syncthreads();  // table in place
  const uint32_t copy = t & (L2_COPIES - 1);
  uint64_t h = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint32_t w[4] = {wm[j].x, wm[j].y, wm[j].z, wm[j].w};
#pragma unroll
    for (int d = 0; d < 4; d++)
#pragma unroll
      for (int b = 0; b < 4; b++) h = (h << 1) + gear_lookup<L2_COPIES>(s_gear, (w[d] >> (8 * b)) & 0xFF, copy);
  }
#else
  // stage the tile: 16 B per lane, coalesced, into the padded layout
#pragma unroll
  for (int k = 0; k < L2_TILE / 16 / L2_NT; k++) {
    const uint32_t i = t + k * L2_NT;          // piece index
    const uint32_t lo = i * 16;                // logical offset in tile
    const uint64_t g = t0 + lo;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (g + 16 <= n) {
      v = load_u4_unaligned(data + g);
    } else if (g < n) {
      uint8_t tmp[16];
      for (int b = 0; b < 16; b++) tmp[b] = (g + b < n) ? data[g + b] : (uint8_t)0;
      __builtin_memcpy(&v, tmp, 16);
    }
    const uint32_t strip = lo / L2_STRIP, o = lo % L2_STRIP;
    *(uint4*)(s_data + (strip + 1) * L2_PSTRIDE + o) = v;
  }
  if (t < 4) {  // 64-byte halo in front of strip 0
    uint4 v = make_uint4(0, 0, 0, 0);
    if (t0 >= 64) v = load_u4_unaligned(data + t0 - 64 + 16 * t);
    *(uint4*)(s_data + (L2_STRIP - 64) + 16 * t) = v;
  }
  __syncthreads();

  const uint32_t copy = t & (L2_COPIES - 1);
  uint64_t h = 0;
  // warm-up over the 64 bytes in front of this strip: afterwards h is exact
  {
    const uint8_t* prev = s_data + t * L2_PSTRIDE + (L2_STRIP - 64);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint4 v = *(const uint4*)(prev + 16 * j);
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int d = 0; d < 4; d++)
#pragma unroll
        for (int b = 0; b < 4; b++) h = (h << 1) + gear_lookup<L2_COPIES>(s_gear, (w[d] >> (8 * b)) & 0xFF, copy);
    }
  }
#endif
  uint32_t Lm[L2_WORDS], Sm[L2_WORDS];
#ifndef HMSE_L2_DIRECT
  const uint8_t* mine = s_data + (t + 1) * L2_PSTRIDE;
#endif
#pragma unroll
  for (int g = 0; g < L2_WORDS; g++) {
    uint32_t lm = 0, sm = 0;
#pragma unroll
    for (int half = 0; half < 2; half++) {
#ifdef HMSE_L2_DIRECT
      const uint4 v = r[g * 2 + half];
#else
      const uint4 v = *(const uint4*)(mine + g * 32 + half * 16);
#endif
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
      // hit <=> (hi & mask) == 0 <=> hi < 2^(32-bits) (the masks are top bits): one v_min per byte into a
      // running minimum; the rare (1/2048) hits are located by replaying the 16 bytes from the saved state
      const uint64_t h0 = h;
      uint32_t mn = 0xFFFFFFFFu;
#pragma unroll
      for (int d = 0; d < 4; d++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
          h = (h << 1) + gear_lookup<L2_COPIES>(s_gear, (w[d] >> (8 * b)) & 0xFF, copy);
          mn = min(mn, (uint32_t)(h >> 32));
        }
      asm volatile("" : "+v"(mn));  // keep this a single compare per 16 bytes
      if (__builtin_expect(mn < thr_l, 0)) {
        uint64_t hr = h0;
#pragma unroll
        for (int d = 0; d < 4; d++)
#pragma unroll
          for (int b = 0; b < 4; b++) {
            hr = (hr << 1) + gear_lookup<L2_COPIES>(s_gear, (w[d] >> (8 * b)) & 0xFF, copy);
            const uint32_t hi = (uint32_t)(hr >> 32);
            if (hi < thr_l) {
              const uint32_t bit = 1u << (half * 16 + d * 4 + b);
              lm |= bit;
              if (hi < thr_s) sm |= bit;
            }
          }
      }
    }
    Lm[g] = lm; Sm[g] = sm;
  }
  // drop positions at or beyond n
  const uint64_t gs = t0 + (uint64_t)t * L2_STRIP;
  const uint32_t valid = n > gs ? (uint32_t)((n - gs) < (uint64_t)L2_STRIP ? (n - gs) : (uint64_t)L2_STRIP) : 0u;
  uint32_t cnt = 0;
#pragma unroll
  for (int g = 0; g < L2_WORDS; g++) {
    const uint32_t lo = g * 32;
    uint32_t m = valid >= lo + 32 ? 0xFFFFFFFFu : (valid > lo ? ((1u << (valid - lo)) - 1u) : 0u);
    Lm[g] &= m; Sm[g] &= m;
    cnt += __builtin_popcount(Lm[g]);
  }
  uint32_t total;
  uint32_t off = block_exclusive_scan<L2_NT>(cnt, s_red, &total);
  if (t == 0) {
    unsigned long long b = total ? atomicAdd(&hdr->cand_total, (unsigned long long)total) : 0ull;
    s_base = b;
    tinfo[tile].base = b;
    tinfo[tile].count = total;
    if (b + total > cand_cap) atomicOr(&hdr->status, 1u);
  }
  __syncthreads();
  const unsigned long long base = s_base;
#pragma unroll
  for (int g = 0; g < L2_WORDS; g++) {
    uint32_t lm = Lm[g];
    while (lm) {
      const uint32_t j = __builtin_ctz(lm);
      lm &= lm - 1;
      const uint32_t pos = t * L2_STRIP + g * 32 + j;
      const unsigned long long at = base + off++;
      if (at < cand_cap) cand[at] = (pos << 1) | ((Sm[g] >> j) & 1u);
    }
  }
}

// One workgroup per segment.  Fast path: the segment's candidate lists (sorted inside a tile, tiles in
// order) are staged into LDS by all four waves, then wave 0 walks them with ballots.  Segments with
// too many tiles/candidates for LDS (or >= 2 GiB) take the same walk directly from global memory.
constexpr int RS_NT = 256;
constexpr int RS_MAX_TILES = 1024;
constexpr int RS_MAX_CAND = 12288;

__device__ __forceinline__ uint32_t clip_count(unsigned long long base, uint32_t count, uint64_t cap) {
  if (base + count > cap) return base < cap ? (uint32_t)(cap - base) : 0u;
  return count;
}

__global__ __launch_bounds__(RS_NT) void l2_resolve_kernel(const uint64_t* __restrict__ seg_off, uint32_t n_seg, uint64_t n,
                                                            const TileInfo* __restrict__ tinfo,
                                                            const uint32_t* __restrict__ cand, uint64_t cand_cap,
                                                            uint32_t MIN, uint32_t AVG, uint32_t MAX,
                                                            uint64_t* __restrict__ seg_cuts, uint32_t* __restrict__ seg_cnt) {
  __shared__ uint32_t s_tpre[RS_MAX_TILES + 1];
  __shared__ uint32_t s_cand[RS_MAX_CAND];
  __shared__ uint32_t s_red[RS_NT / 64 + 1];
  const uint32_t seg = blockIdx.x;
  if (seg >= n_seg) return;
  const uint32_t t = threadIdx.x, lane = lane_id();
  uint64_t a = seg_off[seg], b = seg_off[seg + 1];
  if (b > n) b = n;
  if (a > b) a = b;
  uint64_t* out = seg_cuts + (a / MIN + seg);
  if (a >= b) { if (t == 0) seg_cnt[seg] = 0; return; }
  const uint64_t first_tile = a / L2_TILE, last_tile = (b - 1) / L2_TILE;
  const uint64_t nt64 = last_tile - first_tile + 1;
  bool staged = nt64 <= RS_MAX_TILES && (b - a) < (1ull << 30);
  uint32_t C = 0;
  if (staged) {
    const uint32_t nt = (uint32_t)nt64;
    uint32_t run = 0;
    for (uint32_t base = 0; base < nt; base += RS_NT) {  // prefix of per-tile counts
      const uint32_t i = base + t;
      uint32_t v = 0;
      if (i < nt) v = clip_count(tinfo[first_tile + i].base, tinfo[first_tile + i].count, cand_cap);
      uint32_t total;
      const uint32_t ex = block_exclusive_scan<RS_NT>(v, s_red, &total);
      if (i < nt) s_tpre[i] = run + ex;
      run += total;
    }
    if (t == 0) s_tpre[nt] = run;
    __syncthreads();
    C = s_tpre[nt];
    staged = C <= RS_MAX_CAND;
    if (staged) {
      // one wave per tile: copy its list, rebased to segment-relative cut offsets (o = pos+1-a)
      for (uint32_t i = t >> 6; i < nt; i += RS_NT / 64) {
        const uint32_t p0 = s_tpre[i], cnt = s_tpre[i + 1] - p0;
        const unsigned long long gb = tinfo[first_tile + i].base;
        const int64_t rel = (int64_t)((first_tile + i) * (uint64_t)L2_TILE + 1) - (int64_t)a;
        for (uint32_t k = lane; k < cnt; k += 64) {
          const uint32_t c = cand[gb + k];
          const int64_t o = rel + (int64_t)(c >> 1);
          // candidates in front of the segment can never be cuts: store offset 0 (always < MIN)
          s_cand[p0 + k] = o > 0 ? (((uint32_t)o << 1) | (c & 1u)) : 0u;
        }
      }
    }
    __syncthreads();
  }
  if (t >= 64) return;  // wave 0 resolves
  uint32_t cnt = 0;
  if (staged) {
    const uint32_t len = (uint32_t)(b - a);
    uint32_t s = 0, idx = 0;
    while (s < len) {
      const uint32_t limit = (len - s > MAX) ? s + MAX : len;
      const uint32_t lo = s + MIN;
      uint32_t cut = limit;
      while (idx < C) {
        const uint32_t k = idx + lane;
        const bool valid = k < C;
        const uint32_t c = valid ? s_cand[k] : 0u;
        const uint32_t o = c >> 1;
        const bool beyond = valid && o >= limit;
        const bool qual = valid && o >= lo && o < limit && ((c & 1u) || (o - s) >= AVG);
        const uint64_t qm = __ballot(qual), bm = __ballot(beyond);
        const uint32_t fq = qm ? (uint32_t)__builtin_ctzll(qm) : 64u;
        const uint32_t fb = bm ? (uint32_t)__builtin_ctzll(bm) : 64u;
        if (fq < fb) { cut = __shfl(o, fq, 64); idx += fq + 1; break; }
        if (bm) { idx += fb; break; }
        idx += 64;
      }
      if (lane == 0) out[cnt] = a + cut;
      cnt++;
      s = cut;
    }
  } else {
    uint64_t s = a;
    uint64_t tile = first_tile;
    uint32_t j = 0;
    unsigned long long tbase = tinfo[tile].base;
    uint32_t tcount = clip_count(tbase, tinfo[tile].count, cand_cap);
    while (s < b) {
      const uint64_t limit = (b - s > MAX) ? s + MAX : b;
      const uint64_t lo = s + MIN;
      uint64_t cut = limit;
      for (;;) {
        while (j >= tcount && tile < last_tile) {
          tile++; j = 0;
          tbase = tinfo[tile].base;
          tcount = clip_count(tbase, tinfo[tile].count, cand_cap);
        }
        if (j >= tcount) break;                          // candidates exhausted for this segment
        if (tile * (uint64_t)L2_TILE >= limit) break;    // whole tile lies beyond the forced cut
        const uint32_t k = j + lane;
        const bool valid = k < tcount;
        const uint32_t c = valid ? cand[tbase + k] : 0u;
        const uint64_t o = tile * (uint64_t)L2_TILE + (c >> 1) + 1;
        const bool beyond = valid && o >= limit;
        const bool qual = valid && o >= lo && o < limit && ((c & 1u) || (o - s) >= AVG);
        const uint64_t qm = __ballot(qual), bm = __ballot(beyond);
        const uint32_t fq = qm ? (uint32_t)__builtin_ctzll(qm) : 64u;
        const uint32_t fb = bm ? (uint32_t)__builtin_ctzll(bm) : 64u;
        if (fq < fb) { cut = __shfl(o, fq, 64); j += fq + 1; break; }
        if (bm) { j += fb; break; }
        j += (uint32_t)__builtin_popcountll(__ballot(valid));
      }
      if (lane == 0) out[cnt] = cut;
      cnt++;
      s = cut;
    }
  }
  if (lane == 0) seg_cnt[seg] = cnt;
}

// exclusive scan of seg_cnt -> seg_base, total -> n_cuts  (single workgroup)
__global__ __launch_bounds__(1024) void l2_scan_kernel(const uint32_t* __restrict__ seg_cnt, uint32_t n_seg,
                                                        uint64_t* __restrict__ seg_base, uint64_t* __restrict__ cuts,
                                                        uint64_t cuts_cap, uint64_t* __restrict__ n_cuts, L2Header* hdr) {
  __shared__ uint32_t s_red[1024 / 64 + 1];
  __shared__ unsigned long long s_run;
  if (threadIdx.x == 0) s_run = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n_seg; base += 1024) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < n_seg ? seg_cnt[i] : 0u;
    uint32_t total;
    const uint32_t ex = block_exclusive_scan<1024>(v, s_red, &total);
    const unsigned long long run = s_run;
    if (i < n_seg) seg_base[i] = run + ex;
    __syncthreads();
    if (threadIdx.x == 0) s_run = run + total;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    *n_cuts = s_run;
    if (cuts_cap > 0) cuts[0] = 0;
    if (s_run + 1 > cuts_cap) atomicOr(&hdr->status, 2u);
  }
}

__global__ __launch_bounds__(256) void l2_copy_kernel(const uint64_t* __restrict__ seg_off, uint32_t n_seg, uint64_t n, uint32_t MIN,
                                                       const uint64_t* __restrict__ seg_cuts, const uint32_t* __restrict__ seg_cnt,
                                                       const uint64_t* __restrict__ seg_base, uint64_t* __restrict__ cuts,
                                                       uint64_t cuts_cap) {
  const uint32_t seg = blockIdx.x;
  if (seg >= n_seg) return;
  uint64_t a = seg_off[seg], b = seg_off[seg + 1];
  if (b > n) b = n;
  if (a > b) a = b;
  const uint64_t* src = seg_cuts + (a / MIN + seg);
  const uint32_t c = seg_cnt[seg];
  const uint64_t dst = 1 + seg_base[seg];
  for (uint32_t i = threadIdx.x; i < c; i += blockDim.x)
    if (dst + i < cuts_cap) cuts[dst + i] = src[i];
}

__global__ void l2_publish_status_kernel(const L2Header* __restrict__ hdr, uint32_t* __restrict__ status) { *status = hdr->status; }

// ---- host side -----------------------------------------------------------------------------------

static uint64_t l2_cand_cap(uint64_t n, const hmse_cfg* cfg) {
  // expected density of easy-mask hits is 2^norm / avg; provision 8x plus slack
  uint64_t per = (uint64_t)cfg->avg_size >> cfg->norm_level;
  if (per < 1) per = 1;
  return n / per * 8 + 65536;
}

struct L2Ws {
  L2Header* hdr; TileInfo* tinfo; uint32_t* cand; uint64_t cand_cap;
  uint32_t* seg_cnt; uint64_t* seg_base; uint64_t* seg_cuts; size_t bytes; bool ok;
};

static L2Ws l2_carve(void* ws, size_t ws_bytes, uint64_t n, uint32_t n_seg, const hmse_cfg* cfg) {
  WsCarver w(ws, ws_bytes);
  L2Ws r;
  const uint64_t n_tiles = (n + L2_TILE - 1) / L2_TILE;
  r.hdr = w.take<L2Header>(1);
  r.tinfo = w.take<TileInfo>(n_tiles + 1);
  r.cand_cap = l2_cand_cap(n, cfg);
  r.cand = w.take<uint32_t>(r.cand_cap);
  r.seg_cnt = w.take<uint32_t>((size_t)n_seg + 1);
  r.seg_base = w.take<uint64_t>((size_t)n_seg + 1);
  r.seg_cuts = w.take<uint64_t>(n / cfg->min_size + n_seg + 2);
  r.bytes = w.off;
  r.ok = w.ok();
  return r;
}

size_t hmse_l2_workspace_bytes_impl(uint64_t n, uint32_t n_seg, const hmse_cfg* cfg) {
  return l2_carve(nullptr, 0, n, n_seg, cfg).bytes;
}

extern "C" int hmse_l2_cdc(const uint8_t* data, uint64_t n, const uint64_t* seg_off, uint32_t n_seg,
                           const hmse_cfg* cfg, uint64_t* cuts, uint64_t cuts_cap, uint64_t* n_cuts,
                           uint32_t* status, void* ws, size_t ws_bytes, void* stream_) {
  return hmse_l2_cdc_impl(data, nullptr, n, seg_off, n_seg, cfg, cuts, cuts_cap, n_cuts, status, ws, ws_bytes, (hipStream_t)stream_);
}

int hmse_l2_cdc_impl(const uint8_t* data, const uint64_t* data_off_dev, uint64_t n, const uint64_t* seg_off, uint32_t n_seg,
                     const hmse_cfg* cfg, uint64_t* cuts, uint64_t cuts_cap, uint64_t* n_cuts, uint32_t* status, void* ws,
                     size_t ws_bytes, hipStream_t stream_) {
  if (hmse_cfg_validate_impl(cfg) != 0) return HMSE_EINVAL;
  if (!cuts || !n_cuts || !status || !seg_off || cuts_cap < 1) return HMSE_EINVAL;
  if (n > 0 && !data) return HMSE_EINVAL;
  if (n_seg == 0 && n > 0) return HMSE_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  (void)hipGetLastError();  // drop stale errors of earlier runtime calls made by the host process
  L2Ws w = l2_carve(ws, ws_bytes, n, n_seg, cfg);
  if (!w.ok) return HMSE_ENOSPC;
  HMSE_FILL(w.hdr, 0, sizeof(L2Header), stream);
  uint32_t ms_hi, ml_hi;
  hmse_cdc_masks_hi(cfg, &ms_hi, &ml_hi);
  const uint64_t n_tiles = (n + L2_TILE - 1) / L2_TILE;
  if (n_tiles > 0x7FFFFFFFull) return HMSE_EINVAL;
  if (n_tiles) {
    PROF_BEGIN(HMSE_STAGE_L2_CDC, stream);
  l2_hash_kernel<<<dim3((uint32_t)n_tiles), dim3(L2_NT), 0, stream>>>(data, data_off_dev, n, ~ml_hi + 1u, ms_hi == 0xFFFFFFFFu ? 1u : ~ms_hi + 1u, w.cand, w.cand_cap, w.tinfo, w.hdr);
  PROF_END(HMSE_STAGE_L2_CDC, stream);
    HMSE_LAUNCH_CHECK();
  }
  if (n_seg) {
    l2_resolve_kernel<<<dim3(n_seg), dim3(RS_NT), 0, stream>>>(seg_off, n_seg, n, w.tinfo, w.cand, w.cand_cap, cfg->min_size,
                                                            cfg->avg_size, cfg->max_size, w.seg_cuts, w.seg_cnt);
    HMSE_LAUNCH_CHECK();
  }
  l2_scan_kernel<<<dim3(1), dim3(1024), 0, stream>>>(w.seg_cnt, n_seg, w.seg_base, cuts, cuts_cap, n_cuts, w.hdr);
  HMSE_LAUNCH_CHECK();
  if (n_seg) {
    l2_copy_kernel<<<dim3(n_seg), dim3(256), 0, stream>>>(seg_off, n_seg, n, cfg->min_size, w.seg_cuts, w.seg_cnt, w.seg_base,
                                                          cuts, cuts_cap);
    HMSE_LAUNCH_CHECK();
  }
  l2_publish_status_kernel<<<dim3(1), dim3(1), 0, stream>>>(w.hdr, status);   // (a kernel, not a memcpy node: common.h)
  HMSE_LAUNCH_CHECK();
  return HMSE_OK;
}
