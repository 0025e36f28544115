// Micro-benchmark: random 16-byte gathers from a table in HBM/L2 — one independent address per lane, 8 in flight per
// lane, 32 wavefronts per CU — against the table size (gfx950).  Prices a global memo table keyed by a hash.
// hipcc -O3 --offload-arch=gfx950 gmem_gather.hip -o gmem_gather && ./gmem_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define IT 128
template <int BYTES>
__global__ __launch_bounds__(1024) void k(const uint8_t* tab, uint32_t mask, uint32_t* sink, uint32_t zipf) {
  uint32_t x = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u, acc = 0;
  for (int it = 0; it < IT; it++) {
    u32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      x = x * 1664525u + 1013904223u;
      uint32_t r = x >> 4;
      if (zipf) { const uint32_t s = (x >> 27) & 15u; r >>= s; }   // skewed: half the accesses fall into 1/2, 1/4, ... of the table
      const size_t off = (size_t)(r & mask) * BYTES;
      if (BYTES == 16) v[j] = *(const u32x4*)(tab + off);
      else { v[j].x = *(const uint32_t*)(tab + off); v[j].y = v[j].z = v[j].w = 0; }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) acc ^= v[j].x ^ v[j].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
int main() {
  uint8_t* tab; uint32_t* sink;
  const size_t maxb = (size_t)2 << 30;
  hipMalloc(&tab, maxb); hipMalloc(&sink, 64); hipMemset(tab, 1, maxb);
  for (int zipf = 0; zipf < 2; zipf++)
    for (size_t bytes : {(size_t)1 << 20, (size_t)4 << 20, (size_t)16 << 20, (size_t)64 << 20, (size_t)256 << 20, (size_t)2 << 30}) {
      const uint32_t mask = (uint32_t)(bytes / 16 - 1);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      k<16><<<512, 1024>>>(tab, mask, sink, zipf);
      hipEventRecord(e0);
      k<16><<<512, 1024>>>(tab, mask, sink, zipf);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double n = 512.0 * 1024 * IT * 8;
      printf("table %6zu MiB %s: %7.3f ms for %.0f M gathers of 16 B = %6.1f G gathers/s  (%.2f cycles @2.4GHz per lane per CU)\n", bytes >> 20,
             zipf ? "skewed " : "uniform", ms, n / 1e6, n / ms / 1e6, ms * 1e-3 * 2.4e9 * 256 / n);
    }
  return 0;
}
