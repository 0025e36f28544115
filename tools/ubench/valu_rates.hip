// Micro-benchmark: issue cost of the integer VALU instructions the MinHash kernel is made of (gfx950).
// Each wave runs CH independent dependency chains of one instruction; 8 waves per SIMD; cycles per wave-instruction
// = elapsed shader clocks * (waves that share a SIMD)^-1 ... reported simply as ns per (wave-instruction) per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CH 8
#define IT 4096
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t a, uint32_t b) {
  uint32_t v[CH];
#pragma unroll
  for (int i = 0; i < CH; i++) v[i] = threadIdx.x * 2654435761u + i * 97u + a;
  for (int it = 0; it < IT; it++) {
#pragma unroll
    for (int i = 0; i < CH; i++) {
      if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
      if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
      if (OP == 2) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[i]) : "v"(b));
      if (OP == 3) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(v[i]) : "v"(b));
      if (OP == 4) asm volatile("v_xor_b32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(v[i]));
      if (OP == 5) asm volatile("v_lshl_add_u32 %0, %0, 2, %0" : "+v"(v[i]));
      if (OP == 6) { uint64_t t; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(t) : "v"(v[i]), "v"(b), "v"((uint64_t)a) : "vcc"); v[i] = (uint32_t)t; }
      if (OP == 7) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
      if (OP == 8) asm volatile("v_min_u32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
      if (OP == 9) asm volatile("v_alignbit_b32 %0, %0, %0, 19" : "+v"(v[i]));
      if (OP == 10) asm volatile("v_bitop3_b32 %0, %0, %1, %0 bitop3:0x96" : "+v"(v[i]) : "v"(b));
      if (OP == 11) asm volatile("v_min3_u32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(b));
      if (OP == 12) asm volatile("v_lshrrev_b32 %0, 13, %0" : "+v"(v[i]));
      if (OP == 13) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(v[i]));
      if (OP == 14) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
      if (OP == 15) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
      if (OP == 16) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(0x07060302u));
      if (OP == 17) asm volatile("v_bfe_u32 %0, %0, 3, 20" : "+v"(v[i]));
      if (OP == 18) asm volatile("v_xad_u32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(b));
      if (OP == 19) asm volatile("v_add3_u32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(b));
      if (OP == 20) asm volatile("v_and_or_b32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(b));
      if (OP == 21) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
      if (OP == 22) asm volatile("v_or_b32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
      if (OP == 23) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(v[i]) : "v"(b));
      if (OP == 24) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(b));
      if (OP == 25) asm volatile("v_mov_b32 %0, %1" : "+v"(v[i]) : "v"(b));
      if (OP == 26) asm volatile("v_max_u32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
      if (OP == 27) asm volatile("v_ashrrev_i32 %0, 5, %0" : "+v"(v[i]));
    }
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < CH; i++) s ^= v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP>
void run(const char* name, uint32_t* d) {
  const int blocks = 256 * 8;  // 8 workgroups of 4 waves per CU = 8 waves per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<blocks, 256>>>(d, 3, 0x85ebca6bu); hipDeviceSynchronize();
  hipEventRecord(e0); k<OP><<<blocks, 256>>>(d, 3, 0x85ebca6bu); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // wave-instructions per SIMD = 8 waves * IT * CH
  const double per = ms * 1e6 / (8.0 * IT * CH);
  printf("%-16s %8.3f ms   %6.2f ns per wave-instruction per SIMD  (= %5.2f clk @2.4GHz)\n", name, ms, per, per * 2.4);
}
int main() {
  uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  run<0>("v_add_u32", d); run<1>("v_mul_lo_u32", d); run<2>("v_mul_u32_u24", d); run<3>("v_mad_u32_u24", d);
  run<4>("v_xor_sdwa", d); run<5>("v_lshl_add_u32", d); run<6>("v_mad_u64_u32", d); run<7>("v_mul_hi_u32", d);
  run<8>("v_min_u32", d); run<9>("v_alignbit_b32", d); run<10>("v_bitop3_b32", d); run<11>("v_min3_u32", d);
  run<12>("v_lshrrev_b32", d); run<13>("v_lshlrev_b32", d); run<14>("v_xor_b32", d); run<15>("v_and_b32", d);
  run<16>("v_perm_b32", d); run<17>("v_bfe_u32", d); run<18>("v_xad_u32", d); run<19>("v_add3_u32", d);
  run<20>("v_and_or_b32", d); run<21>("v_sub_u32", d); run<22>("v_or_b32", d); run<23>("v_lshl_or_b32", d);
  run<24>("v_cndmask_b32", d); run<25>("v_mov_b32", d); run<26>("v_max_u32", d); run<27>("v_ashrrev_i32", d);
  return 0;
}
