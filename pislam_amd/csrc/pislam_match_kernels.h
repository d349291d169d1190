// Brute-force Hamming matching of binary descriptors (README.md:125-128 names
// matching as the consumer of the front-end's output — the reference ships no matcher, so the
// semantics are this library's own: include/pislam_hip.h).
//
// For every query descriptor: the train descriptor with the smallest Hamming distance (ties -> the
// smallest train index) and the smallest distance among all OTHER train descriptors (for a ratio
// test).  One lane = one query held in registers; the train descriptor of an iteration is the same
// for the whole wave, so it is fetched with SCALAR loads (s_load_dwordx8) and XOR-ed as an SGPR
// operand; popcount-accumulate is one instruction (v_bcnt_u32_b32).  best / second are tracked on
// the packed key dist << 16 | index with one max and two mins.  Popcount (VALU issue) bound:
// ~21 instructions per (query, train) pair per wave of 64 queries.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pm {

constexpr int MAX_WORDS = 16;

// grid (ceil(max_q / 256), batch); descriptors [batch][stride_q|stride_t][WORDS]; counts may be null
// (then nq / nt apply to every pair).  Train sets larger than 65535 are rejected by the host.
template <int WORDS>
__global__ __launch_bounds__(256) void k_match(const uint32_t *__restrict__ q, const uint32_t *__restrict__ qcount,
                                               size_t q_stride, uint32_t nq_all, const uint32_t *__restrict__ t,
                                               const uint32_t *__restrict__ tcount, size_t t_stride, uint32_t nt_all,
                                               uint32_t cap_q, uint32_t cap_t, int32_t *__restrict__ idx,
                                               uint32_t *__restrict__ dist, uint32_t *__restrict__ dist2,
                                               size_t out_stride) {
  const int b = blockIdx.y;
  const uint32_t nq = min(qcount ? qcount[b] : nq_all, cap_q);
  const uint32_t nt = min(tcount ? tcount[b] : nt_all, cap_t);
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (blockIdx.x * 256u >= nq) return;                   // whole workgroup idle
  const uint32_t *qp = q + (size_t)b * q_stride + (size_t)min(i, nq - 1) * WORDS;
  uint32_t qd[WORDS];
#pragma unroll
  for (int k = 0; k < WORDS; k++) qd[k] = qp[k];
  const uint32_t *tp = t + (size_t)b * t_stride;
  uint32_t best = 0xffffffffu, second = 0xffffffffu;
  for (uint32_t j = 0; j < nt; j++) {
    const uint32_t *td = tp + (size_t)j * WORDS;          // wave-uniform address -> scalar loads
    uint32_t d = 0;
#pragma unroll
    for (int k = 0; k < WORDS; k++) d += (uint32_t)__popc(qd[k] ^ td[k]);
    const uint32_t key = (d << 16) | j;
    second = min(second, max(best, key));
    best = min(best, key);
  }
  if (i < nq) {
    const size_t o = (size_t)b * out_stride + i;
    idx[o] = nt ? (int32_t)(best & 0xffffu) : -1;
    dist[o] = nt ? best >> 16 : 0xffffffffu;
    dist2[o] = nt > 1 ? second >> 16 : 0xffffffffu;
  }
}

}  // namespace pm
