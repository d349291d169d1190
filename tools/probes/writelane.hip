// v_writelane_b32 probe: lane `round` of each half keeps that half's ballot word.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(const uint32_t *in, uint32_t *out_ref, uint32_t *out_wl) {
  const int lane = threadIdx.x, half = lane >> 5, r = lane & 31;
  uint32_t a = 0, b = 0;
#pragma unroll
  for (int round = 0; round < 8; round++) {
    const uint32_t x = in[round * 64 + lane];
    const uint64_t m = __ballot((x & 1u) != 0);
    const uint32_t w = half ? (uint32_t)(m >> 32) : (uint32_t)m;
    if (r == round) a = w;
    const uint32_t mlo = (uint32_t)m, mhi = (uint32_t)(m >> 32);
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(b) : "s"(mlo), "n"(round));
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(b) : "s"(mhi), "n"(32 + round));
  }
  out_ref[lane] = a;
  out_wl[lane] = b;
}
int main() {
  uint32_t h[512], *d, *o1, *o2, r1[64], r2[64];
  for (int i = 0; i < 512; i++) h[i] = (i * 2654435761u) >> 7;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o1, 256); hipMalloc(&o2, 256);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o1, o2);
  hipMemcpy(r1, o1, 256, hipMemcpyDeviceToHost); hipMemcpy(r2, o2, 256, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; i++) if ((i & 31) < 8 && r1[i] != r2[i]) { bad++; printf("lane %d ref %08x wl %08x\n", i, r1[i], r2[i]); }
  printf("mismatches %d\n", bad);
  return 0;
}
