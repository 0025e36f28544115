// Micro-benchmark: what ONE vector memory instruction costs when every lane addresses its own cache line (gfx950).
// The lane-per-stream inflate kernel issues ~10 such instructions per trip (each lane reads its own stream and writes its
// own chunk); this measures ns per instruction per wavefront for each width/alignment, with W wavefronts per CU, one
// instruction + s_waitcnt vmcnt(0) per trip (the kernel's pattern) — and with 4 instructions per wait.
// hipcc -O3 --offload-arch=gfx950 gmem_divergent.hip -o gmem_divergent && ./gmem_divergent
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define IT 2000
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int KIND, int PER>
__global__ __launch_bounds__(64) void k(uint8_t* buf, uint32_t* sink, uint32_t step) {
  extern __shared__ uint32_t pad[];
  uint8_t* r = buf + ((size_t)blockIdx.x * 64 + threadIdx.x) * 9216 + 256;
  uint32_t cur = threadIdx.x * 7u, acc = 0;
  if (threadIdx.x == 999) pad[0] = 1;
  for (int it = 0; it < IT; it++) {
#pragma unroll
    for (int j = 0; j < PER; j++) {
      uint8_t* a = r + cur + j * 1024;
      if (KIND == 0) asm volatile("global_store_byte %0, %1, off" :: "v"(a), "v"(cur) : "memory");
      if (KIND == 1) asm volatile("global_store_dword %0, %1, off" :: "v"((uint8_t*)((uintptr_t)a & ~3ull)), "v"(cur) : "memory");
      if (KIND == 2) asm volatile("global_store_dword %0, %1, off" :: "v"(a), "v"(cur) : "memory");
      if (KIND == 3) { uint64_t v = cur; asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(a), "v"(v) : "memory"); }
      if (KIND == 4) { u32x4 v = {cur, cur, cur, cur}; asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(a), "v"(v) : "memory"); }
      if (KIND == 5) { uint64_t v; asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(a) : "memory"); acc ^= (uint32_t)v; }
      if (KIND == 6) { u32x4 v; asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(a) : "memory"); acc ^= v.x ^ v.w; }
      if (KIND == 7) { uint32_t v; asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"((uint8_t*)((uintptr_t)a & ~3ull)) : "memory"); acc ^= v; }
      if (KIND == 8) { uint32_t v; asm volatile("global_load_ubyte %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(a) : "memory"); acc ^= v; }
      if (KIND == 9) { u32x4 v; asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"((uint8_t*)((uintptr_t)a & ~15ull)) : "memory"); acc ^= v.x ^ v.w; }
      if (KIND == 10) { u32x4 v = {cur, cur, cur, cur}; asm volatile("global_store_dwordx4 %0, %1, off" :: "v"((uint8_t*)((uintptr_t)a & ~15ull)), "v"(v) : "memory"); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    cur += step;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
template <int KIND, int PER>
static void run(const char* name, uint8_t* buf, uint32_t* sink, int wg_per_cu, uint32_t step) {
  const int blocks = 256 * wg_per_cu;
  const size_t lds = wg_per_cu == 4 ? 36864 : wg_per_cu == 8 ? 18432 : wg_per_cu == 16 ? 9216 : 73728;
  hipFuncSetAttribute((const void*)k<KIND, PER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<KIND, PER><<<blocks, 64, lds>>>(buf, sink, step);
  hipEventRecord(e0);
  k<KIND, PER><<<blocks, 64, lds>>>(buf, sink, step);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s %2d waves/CU  %d per wait  step %2u B: %8.1f ns per trip  %7.1f ns per instruction-wave  (%.0f cycles @2.4GHz per CU per instruction)\n",
         name, wg_per_cu, PER, step, ms * 1e6 / IT, ms * 1e6 / IT / PER, ms * 1e6 / IT / PER / wg_per_cu * 2.4);
}
int main() {
  uint8_t* buf; uint32_t* sink;
  const size_t bytes = (size_t)256 * 16 * 64 * 9216 + 65536;
  hipMalloc(&buf, bytes); hipMalloc(&sink, 64); hipMemset(buf, 1, bytes);
  for (int w : {4, 8, 16}) {
    run<0, 1>("store byte", buf, sink, w, 3);
    run<1, 1>("store dword aligned", buf, sink, w, 3);
    run<2, 1>("store dword unaligned", buf, sink, w, 3);
    run<3, 1>("store dwordx2 unaligned", buf, sink, w, 3);
    run<4, 1>("store dwordx4 unaligned", buf, sink, w, 3);
    run<10, 1>("store dwordx4 aligned", buf, sink, w, 16);
    run<8, 1>("load byte", buf, sink, w, 3);
    run<7, 1>("load dword aligned", buf, sink, w, 3);
    run<5, 1>("load dwordx2 unaligned", buf, sink, w, 3);
    run<6, 1>("load dwordx4 unaligned", buf, sink, w, 3);
    run<9, 1>("load dwordx4 aligned", buf, sink, w, 16);
    run<0, 4>("store byte", buf, sink, w, 3);
    run<4, 4>("store dwordx4 unaligned", buf, sink, w, 3);
  }
  return 0;
}
