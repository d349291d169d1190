// Stand-alone timing of pm::k_match_mfma<8> / pm::k_match<8> on random descriptors (256 pairs x 981 x 981).
//   hipcc --offload-arch=gfx950 -O3 -I ../../pislam_amd/csrc [-DMF_ABLATE=n] match_probe.hip -o match_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "pislam_match_kernels.h"
int main(int argc, char **argv) {
  const int batch = 256, n = argc > 1 ? atoi(argv[1]) : 981, stride = 4096, gx = argc > 2 ? atoi(argv[2]) : 32;
  std::vector<uint32_t> h((size_t)batch * stride * 8);
  uint64_t s = 88172645463325252ull;
  for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)(s >> 16); }
  std::vector<uint32_t> cnt(batch, n);
  uint32_t *dq, *dt, *dc, *dd, *d2; int32_t *di;
  hipMalloc(&dq, h.size() * 4); hipMalloc(&dt, h.size() * 4); hipMalloc(&dc, batch * 4);
  hipMalloc(&di, (size_t)batch * stride * 4); hipMalloc(&dd, (size_t)batch * stride * 4); hipMalloc(&d2, (size_t)batch * stride * 4);
  hipMemcpy(dq, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dt, h.data() + 12345, (h.size() - 12345) * 4, hipMemcpyHostToDevice);
  hipMemcpy(dc, cnt.data(), batch * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int variant = 0; variant < 2; variant++) {
    float best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
      hipEventRecord(e0);
      if (variant == 0)
        hipLaunchKernelGGL(pm::k_match_mfma<8>, dim3(gx, batch), dim3(64 * pm::MF_WAVES), 0, 0, dq, dc, (size_t)stride * 8, 0u, dt, dc,
                           (size_t)stride * 8, 0u, (uint32_t)stride, (uint32_t)stride, di, dd, d2, (size_t)stride);
      else
        hipLaunchKernelGGL(pm::k_match<8>, dim3(16, batch), dim3(pm::QPW * pm::SPLIT), 0, 0, dq, dc, (size_t)stride * 8, 0u, dt, dc,
                           (size_t)stride * 8, 0u, (uint32_t)stride, (uint32_t)stride, di, dd, d2, (size_t)stride);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    std::vector<uint32_t> o(16);
    hipMemcpy(o.data(), dd, 64, hipMemcpyDeviceToHost);
    printf("%s: %.3f ms  (dist[0..3] = %u %u %u %u)\n", variant == 0 ? "k_match_mfma<8>" : "k_match<8>     ", best, o[0], o[1], o[2], o[3]);
  }
  return 0;
}
