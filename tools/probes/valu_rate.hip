// Issue-rate probe for a few integer VALU instructions on gfx950 (cycles per wave64 instruction).
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP 64
template <int OP>
__global__ void k(uint32_t *out, int iters) {
  uint32_t a[8];
  for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 2654435761u + i;
  long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < REP / 8; r++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (OP == 0) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (OP == 1) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (OP == 2) asm volatile("v_sad_u8 %0, %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (OP == 3) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (OP == 4) asm volatile("v_perm_b32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (OP == 5) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (OP == 6) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (OP == 7) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (OP == 8) asm volatile("v_min3_u32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (OP == 9) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (OP == 10) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (OP == 11) asm volatile("v_add3_u32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (OP == 12) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (OP == 13) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (OP == 14) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (OP == 15) asm volatile("v_lerp_u8 %0, %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (OP == 16) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (OP == 17) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (OP == 18) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (OP == 19) asm volatile("v_cvt_f32_u32 %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (OP == 20) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (OP == 21) asm volatile("v_bitop3_b32 %0, %0, %1, %0 bitop3:0xc8" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
      }
    }
  }
  long long t1 = clock64();
  uint32_t s = 0;
  for (int i = 0; i < 8; i++) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (uint32_t)(t1 - t0);
}

template <int OP>
void run(const char *name, uint32_t *d) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  // 1 wave per SIMD on every CU (256 CUs x 4 SIMDs): 1024 workgroups of 64 threads
  hipLaunchKernelGGL(k<OP>, dim3(1024), dim3(64), 0, 0, d, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(1024), dim3(64), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  uint32_t cyc;
  hipMemcpy(&cyc, d + (1 << 20), 4, hipMemcpyDeviceToHost);
  printf("%-18s %6.2f clock64-ticks/instr   %.3f ms  (%.2f ns/instr/wave)\n", name, (double)cyc / (iters * REP), ms,
         ms * 1e6 / (iters * (double)REP));
}

int main() {
  uint32_t *d;
  hipMalloc(&d, ((1 << 20) + 16) * 4);
  {
    // full occupancy: 8 waves per SIMD on every CU -> sustained chip-wide issue rate (and clock)
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int wpc : {4, 8, 16, 32}) {
      hipLaunchKernelGGL(k<1>, dim3(256 * wpc), dim3(64), 0, 0, d, 10);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<1>, dim3(256 * wpc), dim3(64), 0, 0, d, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double winstr = 256.0 * wpc * iters * REP;
      printf("v_bcnt, %2d waves/CU: %.3f ms  %.1f G wave-instr/s  (= %.2f GHz x 1024 SIMDs / 4)\n", wpc, ms,
             winstr / ms / 1e6, winstr / ms / 1e6 * 4 / 1024);
    }
  }
  run<0>("v_xor_b32", d);
  run<1>("v_bcnt_u32_b32", d);
  run<2>("v_sad_u8", d);
  run<3>("v_pk_max_u16", d);
  run<4>("v_perm_b32", d);
  run<5>("v_mul_lo_u32", d);
  run<6>("v_dot4_u32_u8", d);
  run<7>("v_alignbyte_b32", d);
  run<8>("v_min3_u32", d);
  run<9>("v_pk_mul_lo_u16", d);
  run<10>("v_mad_u32_u24", d);
  run<11>("v_add3_u32", d);
  run<12>("v_mul_hi_u32", d);
  run<13>("v_mul_u32_u24", d);
  run<14>("v_mul_hi_u32_u24", d);
  run<15>("v_lerp_u8", d);
  run<16>("v_mbcnt_lo", d);
  run<17>("v_pk_add_u16", d);
  run<18>("v_add_u32_sdwa", d);
  run<19>("v_cvt_f32_u32", d);
  run<20>("v_mul_f32", d);
  run<21>("v_bitop3_b32", d);
  return 0;
}
