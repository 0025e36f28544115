// l4_lsh.hip — L4b LSH banding and base selection for gfx950.
//
// Replaces the band tables of README.md:1375-1383, 1937-1945, 1987-1996: the
// signature is split into b bands of r rows; band key = MurmurHash3_x86_32 over the band's r*4
// bytes with seed = band index (bucket = key & (2^band_bits-1)); two chunks are candidates iff a
// WHOLE band is equal, and base[i] = the earliest such chunk before i (or -1).  The reference keeps
// per-bucket id lists on SD; here each band is one open-addressing table in HBM whose slots are
// claimed by band CONTENT (bucket-hash collisions are resolved by comparing the rows), with the
// same order-independent atomicMin rule as L3 dedupe.
#include "common.h"

constexpr uint32_t LS_EMPTY = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t murmur3_mix(uint32_t h, uint32_t k) {
  k *= 0xcc9e2d51u; k = rotl32(k, 15); k *= 0x1b873593u;
  h ^= k; h = rotl32(h, 13); return h * 5u + 0xe6546b64u;
}
__device__ __forceinline__ uint32_t murmur3_words(const uint32_t* p, uint32_t nwords, uint32_t seed) {
  uint32_t h = seed;
  uint32_t i = 0;
  // a band is 16-byte aligned whenever rows is a multiple of 4 (signatures are 512-byte rows of a torch tensor): whole
  // 16-byte loads, so a lane's 128-byte band costs 8 load instructions instead of 32 cache-line touches
  if ((((uintptr_t)p) & 15u) == 0) {
    for (; i + 4 <= nwords; i += 4) {
      const uint4 v = *(const uint4*)(p + i);
      h = murmur3_mix(h, v.x); h = murmur3_mix(h, v.y); h = murmur3_mix(h, v.z); h = murmur3_mix(h, v.w);
    }
  }
  for (; i < nwords; i++) h = murmur3_mix(h, p[i]);
  h ^= nwords * 4u;
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}

__global__ __launch_bounds__(256) void lsh_keys_kernel(const uint32_t* __restrict__ sig, uint64_t i0, uint64_t n, uint32_t bands,
                                                        uint32_t rows, uint32_t* __restrict__ keys, const uint64_t* __restrict__ st = nullptr) {
  if (st) { i0 = st[SB_U_OLD]; n = st[SB_U_NEW]; }
  uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n * bands) return;
  g += i0 * bands;   // chunks [i0, i0 + n)
  const uint64_t i = g / bands;
  const uint32_t b = (uint32_t)(g % bands);
  keys[g] = murmur3_words(sig + i * (uint64_t)(bands * rows) + (uint64_t)b * rows, rows, b);
}

__device__ __forceinline__ bool band_equal(const uint32_t* sig, uint32_t nh, uint32_t rows, uint32_t b, uint32_t x, uint32_t y) {
  const uint32_t* px = sig + (uint64_t)x * nh + (uint64_t)b * rows;
  const uint32_t* py = sig + (uint64_t)y * nh + (uint64_t)b * rows;
  bool eq = true;
  for (uint32_t r = 0; r < rows; r++) eq = eq && (px[r] == py[r]);
  return eq;
}

// one thread per (chunk, band): claim / lower the slot holding this band's content
__global__ __launch_bounds__(256) void lsh_insert_kernel(const uint32_t* __restrict__ sig, const uint32_t* __restrict__ keys,
                                                          uint64_t i0, uint64_t n, uint32_t bands, uint32_t rows, uint32_t* tables,
                                                          uint32_t slots, const uint64_t* __restrict__ st = nullptr) {
  if (st) { i0 = st[SB_U_OLD]; n = st[SB_U_NEW]; }
  uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n * bands) return;
  g += i0 * bands;   // chunks [i0, i0 + n) join tables that may already hold [0, i0)
  const uint32_t i = (uint32_t)(g / bands), b = (uint32_t)(g % bands);
  uint32_t* table = tables + (uint64_t)b * slots;
  const uint32_t mask = slots - 1, nh = bands * rows;
  // keys are well mixed in every bit; use the HIGH bits for the slot so that the low band_bits
  // (the reference's bucket id) are not the only entropy when tables are larger than 2^band_bits
  uint32_t slot = (keys[g] * 0x9E3779B1u) >> 7 & mask;
  for (;;) {
    uint32_t cur = __hip_atomic_load(&table[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == LS_EMPTY) {
      cur = atomicCAS(&table[slot], LS_EMPTY, i);
      if (cur == LS_EMPTY) return;
    }
    if (cur == i) return;
    if (band_equal(sig, nh, rows, b, cur, i)) { atomicMin(&table[slot], i); return; }
    slot = (slot + 1) & mask;
  }
}

// one thread per chunk: earliest chunk sharing any whole band
__global__ __launch_bounds__(256) void lsh_base_kernel(const uint32_t* __restrict__ sig, const uint32_t* __restrict__ keys,
                                                        uint64_t i0, uint64_t n, uint32_t bands, uint32_t rows,
                                                        const uint32_t* __restrict__ tables, uint32_t slots,
                                                        int64_t* __restrict__ base, const uint64_t* __restrict__ st = nullptr) {
  if (st) { i0 = st[SB_U_OLD]; n = st[SB_U_NEW]; }
  const uint64_t i64 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i64 >= n) return;
  const uint32_t i = (uint32_t)(i0 + i64), mask = slots - 1, nh = bands * rows;
  uint32_t best = LS_EMPTY;
  for (uint32_t b = 0; b < bands; b++) {
    const uint32_t* table = tables + (uint64_t)b * slots;
    uint32_t slot = (keys[(uint64_t)i * bands + b] * 0x9E3779B1u) >> 7 & mask;
    for (;;) {
      const uint32_t cur = table[slot];
      if (cur == LS_EMPTY) break;
      if (cur == i || band_equal(sig, nh, rows, b, cur, i)) { if (cur < i && cur < best) best = cur; break; }
      slot = (slot + 1) & mask;
    }
  }
  base[i] = best == LS_EMPTY ? -1 : (int64_t)best;
}

static uint32_t lsh_slots(uint64_t n) {
  uint64_t m = 1024;
  while (m < 2 * n) m <<= 1;
  if (m > (1ull << 24)) m = 1ull << 24;  // slot hash keeps 25 bits
  return (uint32_t)m;
}

size_t hmse_l4_lsh_workspace_bytes_impl(uint64_t n, const hmse_cfg* cfg) {
  return hmse_align_up((size_t)lsh_slots(n) * 4 * cfg->bands, 256);
}

extern "C" int hmse_l4_lsh(const uint32_t* sig, uint64_t n_sel, const hmse_cfg* cfg, uint32_t* band_keys, int64_t* base,
                           void* ws, size_t ws_bytes, void* stream_) {
  if (hmse_cfg_validate_impl(cfg) != 0) return HMSE_EINVAL;
  if (n_sel == 0) return HMSE_OK;
  if (!sig || !band_keys || !base) return HMSE_EINVAL;
  if (n_sel > (1ull << 23)) return HMSE_EINVAL;  // load factor <= 0.5 with 2^24 slots per band
  const uint32_t slots = lsh_slots(n_sel);
  const size_t need = (size_t)slots * 4 * cfg->bands;
  if (!ws || ws_bytes < need) return HMSE_ENOSPC;
  hipStream_t stream = (hipStream_t)stream_;
  (void)hipGetLastError();  // drop stale errors of earlier runtime calls made by the host process
  HMSE_FILL(ws, 0xFF, need, stream);
  const uint64_t nb = n_sel * cfg->bands;
  lsh_keys_kernel<<<dim3((uint32_t)((nb + 255) / 256)), dim3(256), 0, stream>>>(sig, 0, n_sel, cfg->bands, cfg->rows, band_keys);
  HMSE_LAUNCH_CHECK();
  lsh_insert_kernel<<<dim3((uint32_t)((nb + 255) / 256)), dim3(256), 0, stream>>>(sig, band_keys, 0, n_sel, cfg->bands, cfg->rows,
                                                                                 (uint32_t*)ws, slots);
  HMSE_LAUNCH_CHECK();
  lsh_base_kernel<<<dim3((uint32_t)((n_sel + 255) / 256)), dim3(256), 0, stream>>>(sig, band_keys, 0, n_sel, cfg->bands, cfg->rows,
                                                                                   (const uint32_t*)ws, slots, base);
  HMSE_LAUNCH_CHECK();
  return HMSE_OK;
}

// Persistent band tables (README.md:1554-1576 "probe LSH ... insert signature"): chunks
// [n_old, n_old + n_new) of the signature array join tables that already hold [0, n_old); base[] of the new chunks is the
// earliest chunk (old or new) sharing a whole band.  Smaller indices win a slot, so the bases of earlier chunks never
// change and a batch costs O(n_new).  With keys_given != 0 the band keys of the new chunks are already in band_keys (loaded
// from a stored band table) and are not recomputed.
extern "C" uint64_t hmse_l4_lsh_slots(uint64_t capacity_chunks) { return lsh_slots(capacity_chunks); }

extern "C" int hmse_l4_lsh_update(const uint32_t* sig_all, uint64_t n_old, uint64_t n_new, const hmse_cfg* cfg, uint32_t* band_keys,
                                  int64_t* base, uint32_t* tables, uint64_t slots, uint32_t keys_given, void* stream_) {
  if (hmse_cfg_validate_impl(cfg) != 0) return HMSE_EINVAL;
  if (!tables || slots < 1024 || (slots & (slots - 1)) || slots > (1ull << 24)) return HMSE_EINVAL;
  if (2 * (n_old + n_new) > slots) return HMSE_ENOSPC;   // load factor <= 0.5
  hipStream_t stream = (hipStream_t)stream_;
  (void)hipGetLastError();
  if (n_old == 0) HMSE_FILL(tables, 0xFF, (size_t)slots * 4 * cfg->bands, stream);
  if (n_new == 0) return HMSE_OK;
  if (!sig_all || !band_keys) return HMSE_EINVAL;
  const uint64_t nb = n_new * cfg->bands;
  if (!keys_given) {
    lsh_keys_kernel<<<dim3((uint32_t)((nb + 255) / 256)), dim3(256), 0, stream>>>(sig_all, n_old, n_new, cfg->bands, cfg->rows, band_keys);
    HMSE_LAUNCH_CHECK();
  }
  lsh_insert_kernel<<<dim3((uint32_t)((nb + 255) / 256)), dim3(256), 0, stream>>>(sig_all, band_keys, n_old, n_new, cfg->bands, cfg->rows, tables,
                                                                                 (uint32_t)slots);
  HMSE_LAUNCH_CHECK();
  if (base) {
    lsh_base_kernel<<<dim3((uint32_t)((n_new + 255) / 256)), dim3(256), 0, stream>>>(sig_all, band_keys, n_old, n_new, cfg->bands, cfg->rows, tables,
                                                                                     (uint32_t)slots, base);
    HMSE_LAUNCH_CHECK();
  }
  return HMSE_OK;
}

int hmse_l4_lsh_update_dyn(const uint32_t* sig_all, uint32_t* band_keys, int64_t* base_all, uint32_t* tables, uint64_t slots,
                           const uint64_t* st, uint64_t cap_chunks, const hmse_cfg* cfg, hipStream_t stream) {
  if (!sig_all || !band_keys || !base_all || !tables || !st || slots < 1024 || (slots & (slots - 1)) || slots > (1ull << 24)) return HMSE_EINVAL;
  const uint64_t nb = cap_chunks * cfg->bands;
  lsh_keys_kernel<<<dim3((uint32_t)((nb + 255) / 256)), dim3(256), 0, stream>>>(sig_all, 0, 0, cfg->bands, cfg->rows, band_keys, st);
  lsh_insert_kernel<<<dim3((uint32_t)((nb + 255) / 256)), dim3(256), 0, stream>>>(sig_all, band_keys, 0, 0, cfg->bands, cfg->rows, tables, (uint32_t)slots, st);
  lsh_base_kernel<<<dim3((uint32_t)((cap_chunks + 255) / 256)), dim3(256), 0, stream>>>(sig_all, band_keys, 0, 0, cfg->bands, cfg->rows, tables, (uint32_t)slots,
                                                                                       base_all, st);
  HMSE_LAUNCH_CHECK();
  return HMSE_OK;
}
