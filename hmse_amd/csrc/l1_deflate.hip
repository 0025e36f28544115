// l1_deflate.hip — placeholder until the encoder lands (returns HMSE_EINVAL).
#include "common.h"
size_t hmse_l1_deflate_workspace_bytes_impl(uint64_t, const hmse_cfg*) { return 256; }
extern "C" int hmse_l1_deflate(const uint8_t*, uint64_t, const uint64_t*, const uint64_t*, const int64_t*, uint64_t,
                               const hmse_cfg*, uint8_t*, uint64_t, uint64_t*, uint8_t*, uint32_t*, void*, size_t, void*) {
  return HMSE_EINVAL;
}
