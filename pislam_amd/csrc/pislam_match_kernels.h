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

constexpr int QPW = 64;                // queries per workgroup pass (one per lane; the waves split the train set)
constexpr int SPLIT = 8;               // waves per workgroup = interleaved slices of the train set
constexpr int MAX_GRID_X = 16;         // query tiles per pair in flight; a workgroup loops over further tiles

// grid (min(ceil(max_q / 64), 16), batch), 64 * SPLIT threads; descriptors [batch][stride][WORDS];
// counts may be null (then nq / nt apply to every pair).  Train sets larger than 65535 are rejected
// by the host.  All waves of a workgroup hold the same 64 queries (one per lane, in registers) and
// scan interleaved slices of the train set (a front-end batch only has ~1000 queries per pair:
// SPLIT x the waves in flight), two descriptors per iteration; the partial (best, second) pairs are
// merged through LDS.  The train descriptor of an iteration is wave-uniform and is fetched with
// SCALAR loads (s_load_dwordx8, XOR-ed as an SGPR operand).  Measured alternatives for that operand:
// LDS tiles read back with broadcast ds_read_b128 (+11 %), per-lane vector loads of the same address
// (+67 %, bound by the L1 return path).  SMEM returns out of order (only lgkmcnt(0) can wait for it),
// so the loads cannot be software-pipelined inside a wave; the other waves of the SIMD cover them.
template <int WORDS>
__global__ __launch_bounds__(QPW *SPLIT) void k_match(const uint32_t *__restrict__ q, const uint32_t *__restrict__ qcount,
                                                      size_t q_stride, uint32_t nq_all, const uint32_t *__restrict__ t,
                                                      const uint32_t *__restrict__ tcount, size_t t_stride,
                                                      uint32_t nt_all, uint32_t cap_q, uint32_t cap_t,
                                                      int32_t *__restrict__ idx, uint32_t *__restrict__ dist,
                                                      uint32_t *__restrict__ dist2, size_t out_stride) {
  __shared__ uint32_t sh_best[SPLIT][QPW], sh_second[SPLIT][QPW];
  const int b = blockIdx.y;
  const uint32_t nq = min(qcount ? qcount[b] : nq_all, cap_q);
  const uint32_t nt = min(tcount ? tcount[b] : nt_all, cap_t);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t *tp = t + (size_t)b * t_stride;
  for (uint32_t q0 = blockIdx.x * (uint32_t)QPW; q0 < nq; q0 += gridDim.x * (uint32_t)QPW) {
    const uint32_t i = q0 + lane;
    const uint32_t *qp = q + (size_t)b * q_stride + (size_t)min(i, nq - 1) * WORDS;
    uint32_t qd[WORDS];
#pragma unroll
    for (int k = 0; k < WORDS; k++) qd[k] = qp[k];
    uint32_t best = 0xffffffffu, second = 0xffffffffu;
    uint32_t j = wave;
    for (; j + SPLIT < nt; j += 2 * SPLIT) {             // two train descriptors per iteration: j and j + SPLIT
      const uint32_t *t0 = tp + (size_t)j * WORDS, *t1 = t0 + SPLIT * WORDS;   // wave-uniform -> scalar loads
      uint32_t d0 = 0, d1 = 0;
#pragma unroll
      for (int k = 0; k < WORDS; k++) {
        d0 += (uint32_t)__popc(qd[k] ^ t0[k]);
        d1 += (uint32_t)__popc(qd[k] ^ t1[k]);
      }
      const uint32_t k0 = (d0 << 16) | j, k1 = (d1 << 16) | (j + SPLIT);
      second = min(second, max(best, k0));
      best = min(best, k0);
      second = min(second, max(best, k1));
      best = min(best, k1);
    }
    if (j < nt) {
      const uint32_t *t0 = tp + (size_t)j * WORDS;
      uint32_t d0 = 0;
#pragma unroll
      for (int k = 0; k < WORDS; k++) d0 += (uint32_t)__popc(qd[k] ^ t0[k]);
      const uint32_t k0 = (d0 << 16) | j;
      second = min(second, max(best, k0));
      best = min(best, k0);
    }
    sh_best[wave][lane] = best;
    sh_second[wave][lane] = second;
    __syncthreads();
    if (wave == 0 && i < nq) {
#pragma unroll
      for (int w = 1; w < SPLIT; w++) {                  // merge sorted pairs: keys are unique per train index
        const uint32_t ob = sh_best[w][lane], os = sh_second[w][lane];
        second = min(min(second, os), max(best, ob));
        best = min(best, ob);
      }
      const size_t o = (size_t)b * out_stride + i;
      idx[o] = nt ? (int32_t)(best & 0xffffu) : -1;
      dist[o] = nt ? best >> 16 : 0xffffffffu;
      dist2[o] = nt > 1 ? second >> 16 : 0xffffffffu;
    }
    __syncthreads();                                     // sh_best / sh_second are reused by the next tile
  }
}

}  // namespace pm
