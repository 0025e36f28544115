// abi.hip — configuration, error strings and workspace sizing of the C-ABI (include/hmse.h).
#include "common.h"

size_t hmse_l2_workspace_bytes_impl(uint64_t n, uint32_t n_seg, const hmse_cfg* cfg);
size_t hmse_l3_sha256_workspace_bytes_impl(uint64_t n_chunks);
size_t hmse_l3_dedup_workspace_bytes_impl(uint64_t n_chunks);
size_t hmse_l4_minhash_workspace_bytes_impl(uint64_t n_chunks);
size_t hmse_l4_lsh_workspace_bytes_impl(uint64_t n_chunks, const hmse_cfg* cfg);
size_t hmse_l1_deflate_workspace_bytes_impl(uint64_t n_chunks, const hmse_cfg* cfg);
size_t hmse_l1_inflate_workspace_bytes_impl(uint64_t n_chunks);
size_t hmse_manifest_pack_workspace_bytes_impl(uint64_t n_chunks);

extern "C" void hmse_cfg_default(hmse_cfg* c) {
  memset(c, 0, sizeof *c);
  c->struct_size = (uint32_t)sizeof *c;
  c->min_size = 2048;
  c->avg_size = 8192;
  c->max_size = 32768;
  c->norm_level = 2;
  c->seg_size = 4u << 20;
  c->n_hashes = 128;
  c->shingle = 4;
  c->seed_base = 0;
  c->bands = 4;
  c->rows = 32;
  c->band_bits = 16;
  c->level = 9;
  c->chain_depth = 0;
  c->layers = HMSE_LAYER_L1 | HMSE_LAYER_L2 | HMSE_LAYER_L3 | HMSE_LAYER_L4;
  c->delta_max_ratio_pct = 0;
}

static int ilog2_u32(uint32_t v) { int l = 0; while (v > 1) { v >>= 1; l++; } return l; }

int hmse_cfg_validate_impl(const hmse_cfg* c) {
  if (!c || c->struct_size != sizeof(hmse_cfg)) return HMSE_EINVAL;
  if (c->avg_size < 256 || (c->avg_size & (c->avg_size - 1))) return HMSE_EINVAL;
  if (c->min_size < 64 || c->min_size > c->avg_size) return HMSE_EINVAL;
  if (c->max_size < c->avg_size || c->max_size > 32768) return HMSE_EINVAL;
  if (c->norm_level > 4 || ilog2_u32(c->avg_size) + (int)c->norm_level > 32) return HMSE_EINVAL;
  if (ilog2_u32(c->avg_size) - (int)c->norm_level < 1) return HMSE_EINVAL;
  if (c->seg_size < c->max_size) return HMSE_EINVAL;
  if (c->n_hashes != 128 || c->shingle != 4) return HMSE_EINVAL;  // device path is specialised (README.md:2575, 2586)
  if (c->bands == 0 || c->rows == 0 || c->bands * c->rows != c->n_hashes || c->bands > 16) return HMSE_EINVAL;
  if (c->band_bits == 0 || c->band_bits > 32) return HMSE_EINVAL;
  if (c->level > 9) return HMSE_EINVAL;
  if (c->chain_depth > 4096) return HMSE_EINVAL;
  return HMSE_OK;
}

extern "C" int hmse_cfg_validate(const hmse_cfg* c) { return hmse_cfg_validate_impl(c); }

void hmse_cdc_masks_hi(const hmse_cfg* cfg, uint32_t* ms_hi, uint32_t* ml_hi) {
  const int bits = ilog2_u32(cfg->avg_size);
  int bs = bits + (int)cfg->norm_level, bl = bits - (int)cfg->norm_level;
  if (bl < 1) bl = 1;
  if (bs > 32) bs = 32;
  *ms_hi = bs >= 32 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> bs);
  *ml_hi = ~(0xFFFFFFFFu >> bl);
}

uint32_t hmse_deflate_depth(const hmse_cfg* cfg) {
  static const uint32_t tab[10] = {2, 2, 3, 4, 6, 8, 12, 16, 24, 32};
  if (cfg->chain_depth) return cfg->chain_depth;
  return tab[cfg->level > 9 ? 9 : cfg->level];
}

extern "C" int hmse_abi_version(void) { return HMSE_ABI_VERSION; }

extern "C" const char* hmse_strerror(int code) {
  switch (code) {
    case HMSE_OK: return "ok";
    case HMSE_EINVAL: return "invalid argument or unsupported configuration";
    case HMSE_ENOSPC: return "buffer or workspace too small";
    case HMSE_EHIP: return "HIP runtime error";
    default: return "unknown error";
  }
}

extern "C" size_t hmse_workspace_bytes(int stage, uint64_t n, const hmse_cfg* cfg) {
  if (hmse_cfg_validate_impl(cfg) != 0) return 0;
  switch (stage) {
    case HMSE_STAGE_L2_CDC: {
      // default segmentation: ceil(n / seg_size) segments (callers with finer segments pass more)
      uint64_t n_seg = n / cfg->seg_size + 2;
      return hmse_l2_workspace_bytes_impl(n, (uint32_t)n_seg, cfg);
    }
    case HMSE_STAGE_L3_SHA256: return hmse_l3_sha256_workspace_bytes_impl(n);
    case HMSE_STAGE_L3_DEDUP: return hmse_l3_dedup_workspace_bytes_impl(n);
    case HMSE_STAGE_L4_MINHASH: return hmse_l4_minhash_workspace_bytes_impl(n);
    case HMSE_STAGE_L4_LSH: return hmse_l4_lsh_workspace_bytes_impl(n, cfg);
    case HMSE_STAGE_L1_DEFLATE: return hmse_l1_deflate_workspace_bytes_impl(n, cfg);
    case HMSE_STAGE_L1_INFLATE: return hmse_l1_inflate_workspace_bytes_impl(n);
    case HMSE_STAGE_READ_ASSEMBLE: return 256;
    case HMSE_STAGE_MANIFEST_PACK: return hmse_manifest_pack_workspace_bytes_impl(n);
    default: return 0;
  }
}

// ---- diagnostics ---------------------------------------------------------------------------------
// ---- fill kernel (see common.h: the library enqueues kernels only) -------------------------------------------------
__global__ __launch_bounds__(256) void hmse_fill_kernel(uint32_t* __restrict__ p, uint32_t word, uint64_t n_words) {
  const uint64_t n4 = n_words >> 2;
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  const bool al16 = (((uintptr_t)p) & 15u) == 0;
  if (al16) {
    const uint4 v = make_uint4(word, word, word, word);
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) ((uint4*)p)[i] = v;
    for (uint64_t i = (n4 << 2) + (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_words; i += stride) p[i] = word;
  } else {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_words; i += stride) p[i] = word;
  }
}
int hmse_fill_async(void* p, uint32_t byte_value, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return HMSE_OK;
  if (!p || (bytes & 3u) || (((uintptr_t)p) & 3u)) return HMSE_EINVAL;
  const uint32_t b = byte_value & 0xFFu;
  const uint32_t word = b | (b << 8) | (b << 16) | (b << 24);
  const uint64_t n_words = bytes >> 2;
  uint64_t blocks = (n_words / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hmse_fill_kernel<<<dim3((uint32_t)blocks), dim3(256), 0, stream>>>((uint32_t*)p, word, n_words);
  return hipGetLastError() == hipSuccess ? HMSE_OK : HMSE_EHIP;
}

int g_hmse_prof = 0;
namespace {
constexpr int PROF_RING = 64;
struct ProfStage { hipEvent_t e0[PROF_RING], e1[PROF_RING]; int n = 0; bool init = false; double ms = 0; uint64_t launches = 0; };
ProfStage g_ps[32];
void prof_flush(ProfStage& p) {
  for (int i = 0; i < p.n; i++) {
    float t = 0;
    if (hipEventSynchronize(p.e1[i]) == hipSuccess && hipEventElapsedTime(&t, p.e0[i], p.e1[i]) == hipSuccess) { p.ms += t; p.launches++; }
  }
  p.n = 0;
}
}  // namespace
void hmse_prof_begin(int stage, hipStream_t s) {
  ProfStage& p = g_ps[stage & 31];
  if (!p.init) { for (int i = 0; i < PROF_RING; i++) { (void)hipEventCreate(&p.e0[i]); (void)hipEventCreate(&p.e1[i]); } p.init = true; }
  if (p.n == PROF_RING) prof_flush(p);
  (void)hipEventRecord(p.e0[p.n], s);
}
void hmse_prof_end(int stage, hipStream_t s) {
  ProfStage& p = g_ps[stage & 31];
  (void)hipEventRecord(p.e1[p.n], s);
  p.n++;
}
// device-side work counters of the diagnostics (tokens written / read by the DEFLATE kernels, per profile slot): allocated
// by the first hmse_profile_enable(1) — the ONE allocation this library makes, and only on the diagnostics path
unsigned long long* g_hmse_prof_ctr = nullptr;
extern "C" void hmse_profile_enable(int on) {
  if (on && !g_hmse_prof_ctr) {
    if (hipMalloc((void**)&g_hmse_prof_ctr, 32 * sizeof(unsigned long long)) != hipSuccess) g_hmse_prof_ctr = nullptr;
    else (void)hipMemset(g_hmse_prof_ctr, 0, 32 * sizeof(unsigned long long));
  }
  g_hmse_prof = on;
}
extern "C" int hmse_profile_counter(int slot, uint64_t* value, int reset) {
  if (slot < 0 || slot > 31 || !value) return HMSE_EINVAL;
  *value = 0;
  if (!g_hmse_prof_ctr) return HMSE_OK;
  unsigned long long v = 0;
  if (hipMemcpy(&v, g_hmse_prof_ctr + slot, sizeof v, hipMemcpyDeviceToHost) != hipSuccess) return HMSE_EHIP;
  *value = v;
  if (reset) { v = 0; if (hipMemcpy(g_hmse_prof_ctr + slot, &v, sizeof v, hipMemcpyHostToDevice) != hipSuccess) return HMSE_EHIP; }
  return HMSE_OK;
}
extern "C" int hmse_profile_read(int stage, double* total_ms, uint64_t* launches, int reset) {
  if (stage < 0 || stage > 31) return HMSE_EINVAL;
  ProfStage& p = g_ps[stage];
  prof_flush(p);
  if (total_ms) *total_ms = p.ms;
  if (launches) *launches = p.launches;
  if (reset) { p.ms = 0; p.launches = 0; }
  return HMSE_OK;
}
