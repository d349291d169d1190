// probe: does ds_read_b32 / ds_read_b64 at a byte-unaligned LDS address return the expected bytes?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t *out, uint64_t *out64) {
  __shared__ __attribute__((aligned(16))) uint8_t buf[256];
  for (int i = threadIdx.x; i < 256; i += 64) buf[i] = (uint8_t)i;
  __syncthreads();
  const int off = threadIdx.x;                         // 0..63: all alignments
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)(buf) + off) : "memory");
  out[threadIdx.x] = v;
  uint64_t w;
  asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(w) : "v"((uint32_t)(uintptr_t)(buf) + off) : "memory");
  out64[threadIdx.x] = w;
}
int main() {
  uint32_t *d; uint64_t *d64; hipMalloc(&d, 64 * 4); hipMalloc(&d64, 64 * 8);
  k<<<1, 64>>>(d, d64);
  uint32_t h[64]; uint64_t h64[64];
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(h64, d64, sizeof h64, hipMemcpyDeviceToHost);
  int bad = 0, bad64 = 0;
  for (int i = 0; i < 64; i++) {
    uint32_t e = (uint32_t)i | ((i + 1) << 8) | ((i + 2) << 16) | ((uint32_t)(i + 3) << 24);
    if (h[i] != e) { if (bad < 4) printf("b32 off %d got %08x exp %08x\n", i, h[i], e); bad++; }
    uint64_t e64 = 0; for (int b = 0; b < 8; b++) e64 |= (uint64_t)((i + b) & 0xff) << (8 * b);
    if (h64[i] != e64) { if (bad64 < 4) printf("b64 off %d got %016llx exp %016llx\n", i, (unsigned long long)h64[i], (unsigned long long)e64); bad64++; }
  }
  printf("unaligned ds_read_b32: %s (%d bad), ds_read_b64: %s (%d bad)\n", bad ? "WRONG" : "ok", bad, bad64 ? "WRONG" : "ok", bad64);
  return 0;
}
