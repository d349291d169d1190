// pislam_prep_kernels.h — image preparation ("next" tier): the 5x5 Gaussian and
// the 7/8 and 13/16 bilinear reductions of reference include/Gaussian.h / include/Bilinear.h.
// The expected arithmetic is the one the reference's own tests state as scalar references
// (test/GaussianTest.cpp:159-215, test/BilinearTest.cpp:171-233); these kernels are bit-exact to it.
// All three are pure HBM streaming kernels (1 byte read + ~1 byte written per pixel).
#pragma once
#include "pislam_dev.h"

namespace pp {

// rounding halving add, GaussianTest.cpp:32 / the vrhadd tree of Gaussian.h:182-237
__device__ __forceinline__ uint32_t rhadd(uint32_t a, uint32_t b) { return (a + b + 1) >> 1; }
// [1 4 6 4 1]/16 as the reference builds it: RHADD(RHADD(RHADD(RHADD(a,e),c),c), RHADD(b,d))
__device__ __forceinline__ uint32_t tap5(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t e) {
  return rhadd(rhadd(rhadd(rhadd(a, e), c), c), rhadd(b, d));
}
// reflect-101 index (GaussianTest.cpp:164-175,191-202): -1 -> 1, -2 -> 2, n -> n-2, n+1 -> n-3
__device__ __forceinline__ int reflect101(int i, int n) {
  i = i < 0 ? -i : i;
  return i >= n ? 2 * (n - 1) - i : i;
}

// ---------------------------------------------------------------------------
// gaussian5x5 — reference Gaussian.h:48 (behaviour: GaussianTest.cpp:159-215).
// grid (tiles_x, tiles_y, batch); 256 threads; output tile 128 x 16.  The tile plus a 2 px halo is
// staged into LDS with the border reflection applied, the vertical pass runs LDS -> LDS, the
// horizontal pass produces 4 pixels per lane and stores them as one dword.
// src and dst must not alias (the ABI stages in-place calls through a temporary).
// ---------------------------------------------------------------------------
constexpr int G_TW = 128, G_TH = 32;
constexpr int G_DW = 40;                             // LDS row pitch in dwords: [3] = left halo dword (x0-4..x0-1),
                                                     // [4..35] = the tile's 128 columns (16-byte aligned), [36] = right halo
constexpr int G_NC = G_TW / 4 + 2;                   // 34 dword columns take part in the vertical pass

// byte-wise rounding halving add of 4 packed pixels in ONE instruction: v_lerp_u8 computes
// (a + b + (c & 1)) >> 1 per byte, i.e. NEON's vrhadd.u8 with c = 0x01010101
// (the SWAR form (a|b) - (((a^b) & 0xfe..) >> 1) takes four; the kernel was VALU-bound on it: 26 M
// wave-instructions per 64 720p frames)
__device__ __forceinline__ uint32_t rhadd4(uint32_t a, uint32_t b) { return __builtin_amdgcn_lerp(a, b, 0x01010101u); }
__device__ __forceinline__ uint32_t tap5x4(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t e) {
  return rhadd4(rhadd4(rhadd4(rhadd4(a, e), c), c), rhadd4(b, d));
}

typedef uint32_t g_u32x4 __attribute__((ext_vector_type(4)));

// One workgroup = one 128 x 32 output tile (36 staged rows: 12 % halo).  Full, 16-byte aligned tiles stage
// their 128 columns with 16-byte loads (8 per row) plus one halo dword on either side; partial or
// unaligned tiles stage dword by dword (aligned loads inside the image, per-byte with the reflect-101
// rule across its border).  Both passes work on 4 packed pixels per lane, the vertical one with a fixed
// dword column per lane (no index arithmetic in its loop), the horizontal taps come from v_alignbyte on
// aligned LDS dwords, and each lane stores one dword.
__global__ __launch_bounds__(256) void k_gaussian5x5(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                     int vstep_src, int vstep_dst, size_t stride_src,
                                                     size_t stride_dst, int width, int height) {
  __shared__ __attribute__((aligned(16))) uint32_t in[(G_TH + 4) * G_DW];
  __shared__ __attribute__((aligned(16))) uint32_t mid[G_TH * G_DW];
  const uint8_t *s = src + (size_t)blockIdx.z * stride_src;
  uint8_t *d = dst + (size_t)blockIdx.z * stride_dst;
  const int x0 = blockIdx.x * G_TW, y0 = blockIdx.y * G_TH;
  const int tid = threadIdx.x;
  const bool aligned4 = (((uintptr_t)s) & 3) == 0 && (vstep_src & 3) == 0;
  const bool vec_ok = (((uintptr_t)s) & 15) == 0 && (vstep_src & 15) == 0 && x0 + G_TW <= width;
  // one dword of the staged tile: aligned load inside the image, per-byte reflect-101 across its border
  auto stage_dword = [&](const uint8_t *row, int gx0) -> uint32_t {
    if (aligned4 && gx0 >= 0 && gx0 + 4 <= width) return *(const uint32_t *)(row + gx0);
    uint32_t w = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int gx = min(max(reflect101(gx0 + k, width), 0), width - 1);
      w |= (uint32_t)row[gx] << (8 * k);
    }
    return w;
  };
  if (vec_ok) {
    // (G_TH + 4) * 8 = 288 vectors of 16 bytes: one per thread, a second one for the first 32 threads, plus the two
    // halo dwords of every row (72 threads).  ALL of a thread's loads are issued before its first LDS store: written
    // as "load; store" per loop iteration the compiler waits for each load before its store, and the workgroup's
    // first barrier then sits behind three memory round trips in a row instead of one.
    static_assert((G_TH + 4) * 8 <= 2 * 256 && (G_TH + 4) * 2 <= 256, "staging assumes <= 2 vectors and 1 halo dword per thread");
    g_u32x4 v0, v1 = (g_u32x4)(0u);
    uint32_t hd = 0;
    const int i1 = tid + 256;
    {
      const int r = tid >> 3, v = tid & 7;
      const int gy = min(max(reflect101(y0 - 2 + r, height), 0), height - 1);
      v0 = *(const g_u32x4 *)(s + (ptrdiff_t)gy * vstep_src + x0 + 16 * v);
    }
    if (i1 < (G_TH + 4) * 8) {
      const int r = i1 >> 3, v = i1 & 7;
      const int gy = min(max(reflect101(y0 - 2 + r, height), 0), height - 1);
      v1 = *(const g_u32x4 *)(s + (ptrdiff_t)gy * vstep_src + x0 + 16 * v);
    }
    if (tid < (G_TH + 4) * 2) {                                // the two halo dwords of every row
      const int r = tid >> 1, side = tid & 1;
      const int gy = min(max(reflect101(y0 - 2 + r, height), 0), height - 1);
      hd = stage_dword(s + (ptrdiff_t)gy * vstep_src, side ? x0 + G_TW : x0 - 4);
    }
    *(g_u32x4 *)&in[(tid >> 3) * G_DW + 4 + 4 * (tid & 7)] = v0;
    if (i1 < (G_TH + 4) * 8) *(g_u32x4 *)&in[(i1 >> 3) * G_DW + 4 + 4 * (i1 & 7)] = v1;
    if (tid < (G_TH + 4) * 2) in[(tid >> 1) * G_DW + ((tid & 1) ? 36 : 3)] = hd;
  } else {
    for (int i = tid; i < (G_TH + 4) * G_NC; i += 256) {
      const int r = i / G_NC, q = i - r * G_NC;
      const int gy = min(max(reflect101(y0 - 2 + r, height), 0), height - 1);
      in[r * G_DW + 3 + q] = stage_dword(s + (ptrdiff_t)gy * vstep_src, x0 - 4 + 4 * q);
    }
  }
  __syncthreads();
  // vertical pass: lane = (row group, dword column), 7 row groups x 34 columns = 238 lanes
  if (tid < 7 * G_NC) {
    const int rg = tid / G_NC, q = tid - rg * G_NC;
    for (int r = rg; r < G_TH; r += 7) {
      const uint32_t *p = in + r * G_DW + 3 + q;
      mid[r * G_DW + 3 + q] = tap5x4(p[0], p[G_DW], p[2 * G_DW], p[3 * G_DW], p[4 * G_DW]);
    }
  }
  __syncthreads();
  // horizontal pass on the vertical result; the staged halo columns already hold the reflected
  // columns (reflection commutes with the column-wise vertical pass), GaussianTest.cpp:189-213
  for (int i = tid; i < G_TH * (G_TW / 4); i += 256) {
    const int r = i >> 5, q = i & 31;
    const int gy = y0 + r, gx = x0 + 4 * q;
    if (gy >= height || gx >= width) continue;
    const uint32_t *p = mid + r * G_DW + 4 + q;        // dword holding columns gx .. gx+3
    const uint32_t wl = p[-1], wc = p[0], wr = p[1];
    const uint32_t o = tap5x4(__builtin_amdgcn_alignbyte(wc, wl, 2), __builtin_amdgcn_alignbyte(wc, wl, 3), wc,
                              __builtin_amdgcn_alignbyte(wr, wc, 1), __builtin_amdgcn_alignbyte(wr, wc, 2));
    uint8_t *out = d + (ptrdiff_t)gy * vstep_dst + gx;
    if (gx + 4 <= width && (((uintptr_t)out) & 3) == 0) {
      *(uint32_t *)out = o;
    } else {
      for (int k = 0; k < 4 && gx + k < width; k++) out[k] = (uint8_t)(o >> (8 * k));
    }
  }
}

// ---------------------------------------------------------------------------
// bilinear7_8 / bilinear13_16 — reference Bilinear.h:42 / :165 (behaviour:
// BilinearTest.cpp:171-196 / :198-233).  One lane per output pixel; every N x N source block
// (N = 8 or 16) yields M x M outputs (M = 7 or 13); the four taps and two filter weights per axis
// come from small constant tables (bilinear_px).  The reference's fixed-point rounding RSHR(a,8) = (a+128)>>8.
// Writes the same full M x M blocks the scalar reference writes (outputs beyond
// floor(w*M/N) x floor(h*M/N) depend on the caller's padding, exactly as in the reference).
// ---------------------------------------------------------------------------
__device__ __forceinline__ int rshr8(int a) { return (a + 128) >> 8; }

template <int N, int M>
__device__ __forceinline__ uint32_t bilinear_px(const uint8_t *__restrict__ s, int vstep_src, int ox, int oy) {
  // filter banks: Bilinear.h:49-52 (7/8) and Bilinear.h:172-180 (13/16; f[10] = 138 as in the reference)
  constexpr int F7[7] = {238, 201, 165, 128, 91, 55, 18};
  constexpr int F13[13] = {226, 167, 108, 49, 246, 187, 128, 69, 10, 207, 138, 89, 30};
  const int bx = ox / M, x = ox - bx * M, by = oy / M, y = oy - by * M;
  int sx = x, sy = y, fx0, fx1, fy0, fy1;
  if (M == 7) {
    fx0 = F7[x]; fx1 = F7[6 - x]; fy0 = F7[y]; fy1 = F7[6 - y];
  } else {
    sx += (x > 3) + (x > 8);      // map13 (BilinearTest.cpp:198-206): skip source columns 4 and 10
    sy += (y > 3) + (y > 8);
    fx0 = F13[x]; fx1 = F13[12 - x]; fy0 = F13[y]; fy1 = F13[12 - y];
  }
  const uint8_t *p = s + (ptrdiff_t)(by * N + sy) * vstep_src + (bx * N + sx);
  const int h0 = rshr8(p[0] * fx0 + p[1] * fx1);
  const int h1 = rshr8(p[vstep_src] * fx0 + p[vstep_src + 1] * fx1);
  return (uint32_t)rshr8(h0 * fy0 + h1 * fy1);
}

// Fast path: one lane = one output row of FOUR horizontally adjacent source blocks: 2 x (4N) source
// bytes come in as 16-byte loads, the 4M outputs leave as M aligned dwords (4M = 28 or 52 bytes,
// and 4 blocks start on a 4-byte boundary); the x loop is unrolled so every filter weight and
// source column is an immediate.  Needs vstep % 16 == 0 and 16-byte aligned bases.
template <int N, int M>
__global__ __launch_bounds__(256) void k_bilinear4(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                   int vstep_src, int vstep_dst, size_t stride_src,
                                                   size_t stride_dst, int width, int height) {
  constexpr int F7[7] = {238, 201, 165, 128, 91, 55, 18};
  constexpr int F13[13] = {226, 167, 108, 49, 246, 187, 128, 69, 10, 207, 138, 89, 30};
  const int nbx = (width + N - 1) / N, nby = (height + N - 1) / N;
  const int nq = (nbx + 3) / 4, oh = nby * M;
  const int t = blockIdx.x * 256 + threadIdx.x;       // flattened (output row, block group): no idle lanes
  const int oy = t / nq, q = t - oy * nq;
  if (oy >= oh) return;
  const int by = oy / M, y = oy - by * M;
  const int sy = (M == 7) ? y : y + (y > 3) + (y > 8);
  int fy0 = 0, fy1 = 0;
#pragma unroll
  for (int k = 0; k < M; k++)
    if (k == y) {
      fy0 = (M == 7) ? F7[k] : F13[k];
      fy1 = (M == 7) ? F7[M - 1 - k] : F13[M - 1 - k];
    }
  const uint8_t *s0 = src + (size_t)blockIdx.z * stride_src + (ptrdiff_t)(by * N + sy) * vstep_src + q * 4 * N;
  uint8_t *d = dst + (size_t)blockIdx.z * stride_dst + (ptrdiff_t)oy * vstep_dst + q * 4 * M;
  constexpr int NV = 4 * N / 16;                     // 16-byte vectors per source row segment
  uint32_t r0[NV * 4], r1[NV * 4];
  const int vmax = ((nbx - 4 * q) * N + 15) / 16;    // vectors that exist (whole blocks only; padding is the caller's)
#pragma unroll
  for (int v = 0; v < NV; v++) {
    uint4 a = make_uint4(0, 0, 0, 0), b = a;
    if (v < vmax) {
      a = *(const uint4 *)(s0 + 16 * v);
      b = *(const uint4 *)(s0 + vstep_src + 16 * v);
    }
    r0[4 * v] = a.x; r0[4 * v + 1] = a.y; r0[4 * v + 2] = a.z; r0[4 * v + 3] = a.w;
    r1[4 * v] = b.x; r1[4 * v + 1] = b.y; r1[4 * v + 2] = b.z; r1[4 * v + 3] = b.w;
  }
  // Horizontal filter: the two taps of an output are adjacent source bytes, so p0*f0 + p1*f1 + 128 is ONE
  // v_dot4_u32_u8 of the source dword with a constant weight word (f0, f1 at the taps' byte positions, zero
  // elsewhere; accumulator 128) — the bytes are never unpacked.  Taps that straddle two dwords take one
  // v_alignbyte first.  Vertical filter: two v_mad_u32_u24.  8 VALU per output (the packed-u16 form took 11;
  // the kernels are VALU-bound: 124 M outputs per 64-frame 720p build).  Every intermediate is what the
  // reference's u16 arithmetic holds: p*f0 + p'*f1 + 128 <= 255*256 + 128 < 65536.
  uint32_t outw[M];
#pragma unroll
  for (int k = 0; k < M; k++) outw[k] = 0;
#pragma unroll
  for (int b = 0; b < 4; b++) {
#pragma unroll
    for (int x = 0; x < M; x++) {
      const int sx = b * N + ((M == 7) ? x : x + (x > 3) + (x > 8));
      const uint32_t fx0 = (M == 7) ? F7[x] : F13[x], fx1 = (M == 7) ? F7[M - 1 - x] : F13[M - 1 - x];
      uint32_t w0, w1, wt;
      if ((sx & 3) == 3) {                             // taps in two dwords (the last group's never are: sx + 1 < 4 N)
        w0 = __builtin_amdgcn_alignbyte(r0[(sx >> 2) + 1], r0[sx >> 2], 3);
        w1 = __builtin_amdgcn_alignbyte(r1[(sx >> 2) + 1], r1[sx >> 2], 3);
        wt = fx0 | (fx1 << 8);
      } else {
        w0 = r0[sx >> 2];
        w1 = r1[sx >> 2];
        wt = (fx0 << (8 * (sx & 3))) | (fx1 << (8 * ((sx & 3) + 1)));
      }
      const uint32_t h0 = __builtin_amdgcn_udot4(w0, wt, 128u, false) >> 8;
      const uint32_t h1 = __builtin_amdgcn_udot4(w1, wt, 128u, false) >> 8;
      const uint32_t o = (__umul24(h0, (uint32_t)fy0) + __umul24(h1, (uint32_t)fy1) + 128u) >> 8;
      const int oi = b * M + x;
      outw[oi >> 2] |= o << (8 * (oi & 3));
    }
  }
  const int nb_here = min(4, nbx - 4 * q);           // blocks that exist in this group
  if (nb_here == 4) {
#pragma unroll
    for (int k = 0; k < M; k++) *(uint32_t *)(d + 4 * k) = outw[k];
  } else {
    for (int i = 0; i < nb_here * M; i++) d[i] = (uint8_t)((outw[i >> 2] >> (8 * (i & 3))) & 0xffu);
  }
}

// Generic path (any alignment): one lane = 4 horizontally adjacent outputs.
template <int N, int M>
__global__ __launch_bounds__(256) void k_bilinear(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                  int vstep_src, int vstep_dst, size_t stride_src,
                                                  size_t stride_dst, int width, int height) {
  const int nbx = (width + N - 1) / N, nby = (height + N - 1) / N;
  const int ow = nbx * M, oh = nby * M;
  const int ox = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, oy = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (ox >= ow || oy >= oh) return;
  const uint8_t *s = src + (size_t)blockIdx.z * stride_src;
  uint8_t *d = dst + (size_t)blockIdx.z * stride_dst + (ptrdiff_t)oy * vstep_dst + ox;
  uint32_t o[4];
#pragma unroll
  for (int k = 0; k < 4; k++) o[k] = ox + k < ow ? bilinear_px<N, M>(s, vstep_src, ox + k, oy) : 0u;
  if (ox + 4 <= ow && (((uintptr_t)d) & 3) == 0) {
    *(uint32_t *)d = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);
  } else {
    for (int k = 0; k < 4 && ox + k < ow; k++) d[k] = (uint8_t)o[k];
  }
}

// ---------------------------------------------------------------------------
// k_zero_margins — the zero padding the pyramid's consumers read, re-established on every build.
// The build rewrites, in level l's slot, the rectangle [0, ww) x [0, wh) (level 0: the blurred frame; level
// l > 0: the whole M x M output blocks of the reduction into it).  What is read OUTSIDE it: the next
// reduction's block padding (up to 15 columns / rows past the level, Bilinear.h:32,155) and FAST's right-edge
// columns (Fast.h:37-40; up to 3 past the level, whole 16-byte vectors when staged).  So a margin of 32 columns
// to the right (rows [0, hA)) and 16 rows below (columns [0, ww + 32)) is zeroed; bytes beyond the margins are
// nobody's input and are left alone.  grid (2 * nlevels, batch): x = 2 l + {0: right, 1: bottom}; one workgroup
// walks its rectangle.
// ---------------------------------------------------------------------------
struct ZeroPlan {
  int nlevels, vstep;
  int row0[16], ww[16], wh[16], slot_rows[16];
};
constexpr int ZM_COLS = 32, ZM_ROWS = 16;
// CHECK = true (debug flag PISLAM_BUILD_CHECK_MARGINS): nothing is written; the non-zero margin bytes are counted
// into *dirty instead — what a caller's PISLAM_BUILD_MARGINS_CLEAN promise is verified with.
template <bool CHECK>
__global__ __launch_bounds__(256) void k_zero_margins(const ZeroPlan Z, uint8_t *__restrict__ pyramids, size_t stride,
                                                      unsigned int *__restrict__ dirty) {
  const int l = blockIdx.x >> 1, bottom = blockIdx.x & 1;
  const int ww = Z.ww[l], wh = Z.wh[l];
  const int rows_all = min(Z.slot_rows[l], wh + ZM_ROWS);
  int x0, x1, r0, r1;                                 // the rectangle [x0, x1) x [r0, r1) inside the slot
  if (!bottom) {
    x0 = ww; x1 = min(Z.vstep, ww + ZM_COLS); r0 = 0; r1 = min(wh, rows_all);
  } else {
    x0 = 0; x1 = min(Z.vstep, ww + ZM_COLS); r0 = min(wh, rows_all); r1 = rows_all;
  }
  if (x1 <= x0 || r1 <= r0) return;
  const int v0 = x0 >> 4, nv = ((x1 + 15) >> 4) - v0;  // 16-byte vectors the rectangle touches per row
  uint8_t *base = pyramids + (size_t)blockIdx.y * stride + (size_t)Z.row0[l] * Z.vstep;
  for (int i = threadIdx.x; i < (r1 - r0) * nv; i += 256) {
    const int r = r0 + i / nv, v = v0 + i % nv;
    uint8_t *p = base + (size_t)r * Z.vstep;
    const int a = max(16 * v, x0), b = min(16 * v + 16, x1);
    if (CHECK) {
      unsigned int nz = 0;
      for (int x = a; x < b; x++) nz += p[x] != 0;
      if (nz) atomicAdd(dirty, nz);
    } else if (a == 16 * v && b == 16 * v + 16 && (((uintptr_t)(p + a)) & 15) == 0) {
      *(g_u32x4 *)(p + a) = (g_u32x4)(0u);
    } else {
      for (int x = a; x < b; x++) p[x] = 0;
    }
  }
}


}  // namespace pp
