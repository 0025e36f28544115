// Micro-benchmark: LDS reads at random byte addresses, natural alignment vs arbitrary alignment (gfx950).
// The DEFLATE matcher compares window bytes at arbitrary offsets; this measures what a wave-instruction of each width
// costs when its per-lane addresses are (a) naturally aligned, (b) off alignment.  32 waves per CU (8 per SIMD), 8 reads
// in flight per wait, 16 KiB window per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define IT 2048
// MODE: 0 = ds_read_b32, 1 = ds_read_b64, 2 = ds_read_b128, 3 = 2 x ds_read_b32 (offset 0,4), 4 = 4 x ds_read_b32
template <int MODE, bool ALIGNED>
__global__ __launch_bounds__(1024) void k(uint32_t* out, uint32_t seed) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  for (uint32_t i = threadIdx.x; i < 16384 / 4 + 16; i += 1024) ((uint32_t*)smem)[i] = i * 2654435761u;
  __syncthreads();
  uint32_t x = threadIdx.x * 2654435761u + seed, acc = 0;
  const uint32_t base = (uint32_t)(uintptr_t)smem;
  constexpr uint32_t W = MODE == 0 ? 4 : MODE == 1 ? 8 : MODE == 2 ? 16 : MODE == 3 ? 8 : 16;
  for (int it = 0; it < IT; it++) {
    uint32_t a[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      x = x * 1664525u + 1013904223u;
      uint32_t off = (x >> 8) & 16383u;
      if (ALIGNED) off &= ~(W - 1u);
      a[j] = base + off;
    }
    if (MODE == 0) {
      uint32_t r[8];
      asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %9\n\tds_read_b32 %2, %10\n\tds_read_b32 %3, %11\n\t"
                   "ds_read_b32 %4, %12\n\tds_read_b32 %5, %13\n\tds_read_b32 %6, %14\n\tds_read_b32 %7, %15\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
                   : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]) : "memory");
#pragma unroll
      for (int j = 0; j < 8; j++) acc ^= r[j];
    } else if (MODE == 1) {
      uint64_t r[8];
      asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %9\n\tds_read_b64 %2, %10\n\tds_read_b64 %3, %11\n\t"
                   "ds_read_b64 %4, %12\n\tds_read_b64 %5, %13\n\tds_read_b64 %6, %14\n\tds_read_b64 %7, %15\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
                   : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]) : "memory");
#pragma unroll
      for (int j = 0; j < 8; j++) acc ^= (uint32_t)r[j] ^ (uint32_t)(r[j] >> 32);
    } else if (MODE == 2) {
      uint4 r[4];
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]) : "memory");
#pragma unroll
      for (int j = 0; j < 4; j++) acc ^= r[j].x ^ r[j].y ^ r[j].z ^ r[j].w;
    } else if (MODE == 3) {
      uint32_t r[8];
      asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:4\n\tds_read_b32 %2, %9\n\tds_read_b32 %3, %9 offset:4\n\t"
                   "ds_read_b32 %4, %10\n\tds_read_b32 %5, %10 offset:4\n\tds_read_b32 %6, %11\n\tds_read_b32 %7, %11 offset:4\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
                   : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]) : "memory");
#pragma unroll
      for (int j = 0; j < 8; j++) acc ^= r[j];
    } else {
      uint32_t r[8];
      asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:4\n\tds_read_b32 %2, %8 offset:8\n\tds_read_b32 %3, %8 offset:12\n\t"
                   "ds_read_b32 %4, %9\n\tds_read_b32 %5, %9 offset:4\n\tds_read_b32 %6, %9 offset:8\n\tds_read_b32 %7, %9 offset:12\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
                   : "v"(a[0]), "v"(a[1]) : "memory");
#pragma unroll
      for (int j = 0; j < 8; j++) acc ^= r[j];
    }
  }
  out[blockIdx.x * 1024 + threadIdx.x] = acc;
}
template <int MODE, bool ALIGNED>
void run(const char* name, uint32_t* d, int bytes_per_group) {
  const int blocks = 256 * 2;  // 2 workgroups of 16 waves per CU = 8 waves per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE, ALIGNED><<<blocks, 1024, 16384 + 64>>>(d, 1); hipDeviceSynchronize();
  hipEventRecord(e0); k<MODE, ALIGNED><<<blocks, 1024, 16384 + 64>>>(d, 2); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // per CU: 32 waves * IT groups, each group = bytes_per_group per lane
  const double clk_per_wave_group = ms * 1e-3 * 2.4e9 / (32.0 * IT);
  printf("%-34s %-9s %8.3f ms  %7.1f CU-clk per wave-group of %3d B/lane  = %5.2f clk per 4 B per wave\n", name, ALIGNED ? "aligned" : "unaligned",
         ms, clk_per_wave_group, bytes_per_group, clk_per_wave_group / (bytes_per_group / 4.0));
}
int main() {
  uint32_t* d; hipMalloc(&d, 512 * 1024 * 4);
  run<0, true>("8 x ds_read_b32 (8 addresses)", d, 32);  run<0, false>("8 x ds_read_b32 (8 addresses)", d, 32);
  run<1, true>("8 x ds_read_b64 (8 addresses)", d, 64);  run<1, false>("8 x ds_read_b64 (8 addresses)", d, 64);
  run<2, true>("4 x ds_read_b128 (4 addresses)", d, 64); run<2, false>("4 x ds_read_b128 (4 addresses)", d, 64);
  run<3, true>("4 x (2 x ds_read_b32) (4 addresses)", d, 32); run<3, false>("4 x (2 x ds_read_b32) (4 addresses)", d, 32);
  run<4, true>("2 x (4 x ds_read_b32) (2 addresses)", d, 32); run<4, false>("2 x (4 x ds_read_b32) (2 addresses)", d, 32);
  return 0;
}
