// Throughput of v_mfma_i32_32x32x32_i8 in chains of 8 dependent instructions (the matcher's tile), with and without
// the 16 table lookups per tile.   hipcc --offload-arch=gfx950 -O3 mfma_i8_rate.hip -o mfma_i8_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256) void k(int *out, int iters, const unsigned *words) {
  __shared__ uint2 tab[256];
  tab[threadIdx.x] = make_uint2(threadIdx.x * 0x01010101u, ~threadIdx.x);
  __syncthreads();
  v4i b[8];
  for (int i = 0; i < 8; i++) b[i] = (v4i){(int)threadIdx.x + i, i, 3 * i, 7};
  unsigned w[8];
  for (int i = 0; i < 8; i++) w[i] = words[(threadIdx.x + i) & 255];
  int best = 0x7fffffff, second = 0x7fffffff;
  for (int it = 0; it < iters; it++) {
    v16i acc = {0};
#pragma unroll
    for (int kk = 0; kk < 8; kk++) {
      v4i a;
      if (MODE & 1) {
        const uint2 e0 = tab[(w[kk] >> (it & 16)) & 255u], e1 = tab[(w[kk] >> ((it & 16) + 8)) & 255u];
        a = (v4i){(int)e0.x, (int)e0.y, (int)e1.x, (int)e1.y};
      } else {
        a = (v4i){(int)w[kk], it, kk, 1};
      }
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b[kk], acc, 0, 0, 0);
    }
    if (MODE & 2) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int lk = r - acc[r] * 65536;
        second = min(second, max(best, lk));
        best = min(best, lk);
      }
    } else {
      best ^= acc[0] ^ acc[5];
    }
    for (int i = 0; i < 8; i++) w[i] += it;
  }
  out[blockIdx.x * 256 + threadIdx.x] = best + second;
}
template <int MODE>
void run(const char *name, int *out, unsigned *words) {
  const int iters = 2000, blocks = 256 * 4;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, words);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, words);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double tiles = (double)blocks * 4 * iters;
  printf("%s: %.3f ms, %.1f ns per wave-tile per SIMD-slot, %.0f TOPS\n", name, ms, ms * 1e6 / (tiles / 1024.0), tiles * 8 * 65536.0 / (ms * 1e-3) / 1e12);
}
int main() {
  int *out; unsigned *words;
  hipMalloc(&out, 256 * 4 * 256 * 4); hipMalloc(&words, 1024); hipMemset(words, 0x5a, 1024);
  run<0>("mfma only            ", out, words);
  run<1>("mfma + table lookups ", out, words);
  run<2>("mfma + epilogue      ", out, words);
  run<3>("mfma + both          ", out, words);
  return 0;
}
