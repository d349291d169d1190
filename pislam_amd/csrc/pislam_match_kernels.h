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
#include <type_traits>

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

// ---------------------------------------------------------------------------
// The same matcher on the matrix cores (gfx950 v_mfma_i32_32x32x32_i8).  All-pairs Hamming distance is a
// GEMM over bits: with the train bits as 0/1 bytes (A, rows) and the query bits as +1/-1 bytes (B, columns),
//   C[j][i] = sum_k t_jk * (2 q_ik - 1) = 2 popc(t_j & q_i) - popc(t_j),   dist(i, j) = popc(q_i) - C[j][i],
// so one 32x32x32 MFMA per descriptor word covers 32 train x 32 query descriptors, and popc(q_i) is a per-lane
// constant (the C layout gives every lane ONE query column: col = lane & 31, rows (reg & 3) + 8 (reg >> 2) +
// 4 (lane >> 5)).  A and B fragments use the same lane -> (index, k-half) rule, so only "row / col = lane & 31,
// the two lane halves split k" is assumed of the operand layout, not the order of k inside a half.
//
// One wave = 32 queries (B fragments of all words stay in registers), sweeping the train set in tiles of 32; a
// tile's raw words are expanded to bytes through a 256-entry LDS table (8 bits -> 8 bytes, two ds_read_b64 per
// word), the next tile's words are loaded while the MFMAs run.  Epilogue per C element: key = row - (C << 16)
// (v_mad_i32_i24), second = med3(best, second, key), best = min(best, key) on tile-local keys; the tile's pair is
// shifted by its first train index and merged into the running pair; popc(q) << 16 is added once at the end.
// Keys order exactly like the VALU kernel's dist << 16 | index, so the results are identical (ties -> the
// smallest train index).  ~3.5 VALU per (query, train) pair per wave instead of ~21: batch 256 x 981 x 981 pairs in
// 0.09 ms against 0.15 ms (tools/probes/match_probe.hip), VALU issue and the MFMA pipe sharing the time evenly.
// ---------------------------------------------------------------------------
typedef int mf_v4i __attribute__((ext_vector_type(4)));
typedef int mf_v16i __attribute__((ext_vector_type(16)));
constexpr int MF_WAVES = 4;            // waves per workgroup: 4 x 32 queries per pass
constexpr int MF_Q = 32 * MF_WAVES;
constexpr int32_t MF_BIG = 0x7ff00000; // "no candidate": larger than any key, small enough to add a train index to


// Pins a prefetched register array at this point of the program: the compiler has to have the loads issued
// before and their values complete here (it otherwise sinks a prefetch down to its first use in the NEXT
// iteration, and every tile then waits out a full global-load latency).
template <int WORDS>
__device__ __forceinline__ void mf_pin(uint32_t (&d)[WORDS]) {
  if constexpr (WORDS == 8)
    asm volatile("" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]));
  else if constexpr (WORDS == 4)
    asm volatile("" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));
  else if constexpr (WORDS == 2)
    asm volatile("" : "+v"(d[0]), "+v"(d[1]));
  else
    asm volatile("" : "+v"(d[0]));
}

template <int WORDS>
__global__ __launch_bounds__(64 * MF_WAVES) void k_match_mfma(const uint32_t *__restrict__ q, const uint32_t *__restrict__ qcount,
                                                              size_t q_stride, uint32_t nq_all, const uint32_t *__restrict__ t,
                                                              const uint32_t *__restrict__ tcount, size_t t_stride,
                                                              uint32_t nt_all, uint32_t cap_q, uint32_t cap_t,
                                                              int32_t *__restrict__ idx, uint32_t *__restrict__ dist,
                                                              uint32_t *__restrict__ dist2, size_t out_stride) {
  __shared__ uint2 tab[256];                             // 8 bits -> 8 bytes of 0 / 1 (bit k -> byte k)
  const int b = blockIdx.y;
  const uint32_t nq = min(qcount ? qcount[b] : nq_all, cap_q);
  if (blockIdx.x * (uint32_t)MF_Q >= nq) return;         // (the grid is sized for the capacity, not for the counts)
  {
    const uint32_t e = threadIdx.x;
    auto spread = [](uint32_t n) {                       // 4 bits -> 4 bytes
      uint32_t v = n | (n << 7);
      v |= v << 14;
      return v & 0x01010101u;
    };
    tab[e] = make_uint2(spread(e & 15u), spread(e >> 4));
  }
  __syncthreads();
  const uint32_t nt = min(tcount ? tcount[b] : nt_all, cap_t);
  const uint32_t lane = threadIdx.x & 63u, col = lane & 31u, half = lane >> 5;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t sh = 16u * half;                        // this lane half's 16 bits of every word
  const uint32_t *tp = t + (size_t)b * t_stride;
  const uint2 *tabp = tab;
  auto expand01 = [&](uint32_t w) -> mf_v4i {            // 16 bits -> 16 bytes of 0 / 1
    const uint32_t w16 = w >> sh;                        // (then the two index bytes are fixed byte selects)
    const uint2 e0 = tabp[w16 & 255u], e1 = tabp[(w16 >> 8) & 255u];
    return (mf_v4i){(int)e0.x, (int)e0.y, (int)e1.x, (int)e1.y};
  };
  auto pm1 = [](int x) -> int { return (int)((((uint32_t)x ^ 0x01010101u) * 0xfeu) | 0x01010101u); };   // 0/1 -> -1/+1 bytes
  for (uint32_t q0 = blockIdx.x * (uint32_t)MF_Q + wave * 32u; q0 < nq; q0 += gridDim.x * (uint32_t)MF_Q) {
    const uint32_t i = q0 + col;
    const uint32_t *qp = q + (size_t)b * q_stride + (size_t)min(i, nq - 1) * WORDS;
    mf_v4i bq[WORDS];
    uint32_t pq = 0;
#pragma unroll
    for (int k = 0; k < WORDS; k++) {
      const uint32_t w = qp[k];
      pq += (uint32_t)__popc(w);
      const mf_v4i e = expand01(w);
      bq[k] = (mf_v4i){pm1(e.x), pm1(e.y), pm1(e.z), pm1(e.w)};
    }
    int32_t gb = MF_BIG, gs = MF_BIG;
    // raw words of the tile in flight: loaded one tile ahead (at the top of an iteration, pinned at its bottom)
    uint32_t tw[WORDS], tn[WORDS];
    if (nt) {
      const uint32_t *t0 = tp + (size_t)min(col, nt - 1) * WORDS;
#pragma unroll
      for (int k = 0; k < WORDS; k++) tn[k] = t0[k];
    }
    // one tile of 32 train descriptors; FULL = all 32 rows exist (the tail tile masks the missing rows)
    auto tile = [&](uint32_t jt, auto full_tag) {
      constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
      for (int k = 0; k < WORDS; k++) tw[k] = tn[k];
      {
        const uint32_t *t1 = tp + (size_t)min(jt + 32u + col, nt - 1) * WORDS;   // (the last tile re-reads a valid row)
#pragma unroll
        for (int k = 0; k < WORDS; k++) tn[k] = t1[k];
      }
      mf_v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < WORDS; k++) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(expand01(tw[k]), bq[k], acc, 0, 0, 0);
      int32_t b2 = MF_BIG, s2 = MF_BIG;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int ofs = (r & 3) + 8 * (r >> 2);          // row of this accumulator register (+ 4 * half)
        int32_t lk = __mul24(acc[r], -65536) + ofs;      // v_mad_i32_i24 (|C| <= 256)
        if (!FULL && jt + (uint32_t)ofs + 4u * half >= nt) lk = MF_BIG;
        asm("v_med3_i32 %0, %1, %2, %3" : "=v"(s2) : "v"(b2), "v"(s2), "v"(lk));   // b2 <= s2: the second smallest of the three
        b2 = min(b2, lk);
      }
      const int32_t jofs = (int32_t)(jt + 4u * half);
      b2 += jofs;
      s2 += jofs;
      gs = min(min(gs, s2), max(gb, b2));
      gb = min(gb, b2);
      mf_pin<WORDS>(tn);
    };
    uint32_t jt = 0;
    for (; jt + 32u <= nt; jt += 32u) tile(jt, std::true_type{});
    if (jt < nt) tile(jt, std::false_type{});
    // the two lane halves hold the even / odd groups of four train rows of the same query
    const int32_t ob = __shfl_xor(gb, 32, 64), os = __shfl_xor(gs, 32, 64);
    gs = min(min(gs, os), max(gb, ob));
    gb = min(gb, ob);
    if (half == 0 && i < nq) {
      const size_t o = (size_t)b * out_stride + i;
      const uint32_t kb = (uint32_t)gb + (pq << 16), ks = (uint32_t)gs + (pq << 16);
      idx[o] = nt ? (int32_t)(kb & 0xffffu) : -1;
      dist[o] = nt ? kb >> 16 : 0xffffffffu;
      dist2[o] = nt > 1 ? ks >> 16 : 0xffffffffu;
    }
  }
}

}  // namespace pm
