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
// (the work of ONE lane: item t = (output row oy, group q of four source blocks) of one image — shared by k_bilinear4,
//  one launch per level, and k_bilinear_chain, all levels of a pyramid build in one launch)
// COH (k_bilinear_chain): the source rows were written by OTHER workgroups of the same grid, on the same XCD (the chain
// keeps a frame on one XCD and checks it: see there).  Their plain stores sit in that XCD's L2, which is the coherence point
// of its CUs; what a consumer must not do is hit its own CU's vector L1, which no other CU's store ever refreshes — so the
// source comes in through 16-byte `sc1` loads (they bypass the L1 and are served by the L2 at the plain rate;
// /opt/skills/guides/MI355X_MICROARCH.md, inter-workgroup visibility).  Measured alternatives, 64 720p frames, seven launches
// = 108 us: ordinary accesses between an agent-scope release per band and an acquire per workgroup 1218 us (12 000 L2
// write-backs and L1 invalidates); agent-scope relaxed atomics on both sides (8-byte loads, 4-byte write-through stores:
// every dword its own fabric write) 1031 us.
__device__ __forceinline__ g_u32x4 load16_sc1(const uint8_t *p) {
  g_u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int N, int M, bool COH = false>
__device__ __forceinline__ void bilinear4_item(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int vstep_src,
                                               int vstep_dst, int width, int height, int t) {
  constexpr int F7[7] = {238, 201, 165, 128, 91, 55, 18};
  constexpr int F13[13] = {226, 167, 108, 49, 246, 187, 128, 69, 10, 207, 138, 89, 30};
  const int nbx = (width + N - 1) / N, nby = (height + N - 1) / N;
  const int nq = (nbx + 3) / 4, oh = nby * M;
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
  const uint8_t *s0 = src + (ptrdiff_t)(by * N + sy) * vstep_src + q * 4 * N;
  uint8_t *d = dst + (ptrdiff_t)oy * vstep_dst + q * 4 * M;
  constexpr int NV = 4 * N / 16;                     // 16-byte vectors per source row segment
  uint32_t r0[NV * 4], r1[NV * 4];
  g_u32x4 ca[NV], cb[NV];
  const int vmax = ((nbx - 4 * q) * N + 15) / 16;    // vectors that exist (whole blocks only; padding is the caller's)
#pragma unroll
  for (int v = 0; v < NV; v++) {
    if (COH) {
      // (asm loads: the compiler does not know their latency — all of them are issued first, ONE wait below covers them)
      g_u32x4 a = (g_u32x4)(0u), b = (g_u32x4)(0u);
      if (v < vmax) {
        a = load16_sc1(s0 + 16 * v);
        b = load16_sc1(s0 + vstep_src + 16 * v);
      }
      ca[v] = a;
      cb[v] = b;
    } else {
      uint4 a = make_uint4(0, 0, 0, 0), b = a;
      if (v < vmax) {
        a = *(const uint4 *)(s0 + 16 * v);
        b = *(const uint4 *)(s0 + vstep_src + 16 * v);
      }
      r0[4 * v] = a.x; r0[4 * v + 1] = a.y; r0[4 * v + 2] = a.z; r0[4 * v + 3] = a.w;
      r1[4 * v] = b.x; r1[4 * v + 1] = b.y; r1[4 * v + 2] = b.z; r1[4 * v + 3] = b.w;
    }
  }
  if (COH) {
    // the wait takes the loaded registers as operands: nothing that uses them can be scheduled in front of it
    if constexpr (NV == 2) {
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(ca[0]), "+v"(ca[1]), "+v"(cb[0]), "+v"(cb[1])::"memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(ca[0]), "+v"(ca[1]), "+v"(ca[2]), "+v"(ca[3]), "+v"(cb[0]), "+v"(cb[1]), "+v"(cb[2]), "+v"(cb[3])::"memory");
    }
#pragma unroll
    for (int v = 0; v < NV; v++) {
      r0[4 * v] = ca[v].x; r0[4 * v + 1] = ca[v].y; r0[4 * v + 2] = ca[v].z; r0[4 * v + 3] = ca[v].w;
      r1[4 * v] = cb[v].x; r1[4 * v + 1] = cb[v].y; r1[4 * v + 2] = cb[v].z; r1[4 * v + 3] = cb[v].w;
    }
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

template <int N, int M>
__global__ __launch_bounds__(256) void k_bilinear4(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                   int vstep_src, int vstep_dst, size_t stride_src,
                                                   size_t stride_dst, int width, int height) {
  // flattened (output row, block group): no idle lanes
  bilinear4_item<N, M>(src + (size_t)blockIdx.z * stride_src, dst + (size_t)blockIdx.z * stride_dst, vstep_src, vstep_dst, width,
                       height, (int)(blockIdx.x * 256 + threadIdx.x));
}

// ---------------------------------------------------------------------------
// k_bilinear_chain — EVERY reduction of a pyramid build (levels 1 .. n-1 of every frame of the batch) in ONE launch.
// Level by level the build was seven dependent launches; the small levels move a few MB each and took ~15 us apiece
// where ~3 us of work was in them: launch floor, ramp-up and tail, seven times.  Here the grid holds the work items of
// all levels, level-major (all of level 1, then all of level 2, ...), and a workgroup of level l + 1 starts as soon as the
// rows of level l it reads are complete — not when the whole level is:
//   * level l's output rows are counted in BANDS of CH_BAND rows per frame: a workgroup that has stored its items adds,
//     per band it touched, the number of items it finished there (all stores acknowledged by the L2, workgroup barrier,
//     then ONE agent-scope add per band by thread 0); a band is complete when its counter holds rows x items-per-row;
//   * a workgroup of level l + 1 computes the source rows its items read, polls those bands' counters (relaxed agent-scope
//     loads, one lane per band), then — behind a workgroup barrier — runs its items (pp::bilinear4_item: the arithmetic of
//     the one-launch-per-level kernel, bit for bit);
//   * ONE XCD PER FRAME.  Per-XCD L2s are not coherent with each other, and the per-workgroup agent-scope release
//     (L2 write-back) / acquire (L1 invalidate) that would make a cross-XCD hand-over visible cost more than the launches
//     they replace (see bilinear4_item<COH>).  So all workgroups of a frame are placed on the same XCD — workgroup b of a
//     level works on frame 8 g + b % 8, and blocks are dealt to XCDs round-robin — where the L2 is the common coherence point:
//     plain stores + `sc1` loads, no cache maintenance at all.  That placement is an observation, not a contract, so it
//     is VERIFIED: the first workgroup of a frame records its HW_REG_XCC_ID, every later one compares; a mismatch raises
//     the fault flag (the library reports it at the next call and builds one launch per level from then on);
//   * forward progress: workgroups are dispatched in blockIdx order and a workgroup only ever waits for workgroups with
//     SMALLER indices, so the earliest unfinished workgroup is always resident with all its inputs complete.  Nothing
//     documents that order either: the wait is bounded, a workgroup that gives up raises the same sticky flag (counter
//     block + host-mapped fault word), and such a launch does not re-arm;
//   * the launch's last workgroup re-arms the counters: a captured launch replays with no host-side reset.
// Level 0 (the blurred frame) comes from the launch before this one (stream order): level 1 waits for nothing.
// ---------------------------------------------------------------------------
constexpr int CH_BAND = 16;
struct ChainPlan {
  int nlevels, vstep, batch, bands_per_frame;
  int groups;                  // ceil(batch / 8) groups of eight frames (frame = 8 g + b % 8)
  int wg0[17];                 // first workgroup of level l (1 .. nlevels-1), multiples of 8; wg0[nlevels] = grid size
  // (Launch order: level-major.  Ordering by diagonals d = g + l - 1 — group g's level l + 1 one step behind its level l, so
  //  that the small levels of the early groups run beside the big levels of the late ones instead of all at the end — was
  //  measured: 202 us per 64-frame build against 166: consumers then sit right behind their producers in the dispatch order,
  //  and a waiting workgroup holds a slot.)
  int wpf[16];                 // workgroups per frame of level l
  int kind[16];                // the reduction INTO level l: 1 = 7/8, 2 = 13/16
  int row0[16];                // pyramid row of level l
  int sw[16], sh[16];          // width / height of level l - 1 (the source of the reduction into level l)
  int nq[16], oh[16];          // level l: items per output row, output rows written
  int band0[16];               // index (within a frame's block) of level l's first band counter
};
// ctr: [batch][bands_per_frame] band counters (rounded up to a 128-byte line), then — every word on a 128-byte line of its
// own (CH_LINE dwords apart) — [0] sticky fault (a wait timed out / a frame met two XCDs), [1] shards complete,
// [2 .. 2 + CH_SHARDS) workgroups done per shard, [2 + CH_SHARDS ..) the XCD of each frame + 1 (0 = not yet known).  All
// zero between launches (the fault word: until the host has seen it).
// (Measured, 64 720p frames: ONE done counter next to the fault word every poller reads and the per-frame XCD words every
//  workgroup reads made the kernel 837 us; without the done counter 134 us — 12 500 returning atomics on a line that 12 500
//  other accesses want.  Hence the shards and the lines.)
constexpr int CH_LINE = 32, CH_SHARDS = 64;
__host__ __device__ constexpr size_t chain_tail_ofs(size_t batch, size_t bands_per_frame) {
  return (batch * bands_per_frame + CH_LINE - 1) / CH_LINE * CH_LINE;
}
__host__ __device__ constexpr size_t chain_words(size_t batch, size_t bands_per_frame, size_t groups) {
  return chain_tail_ofs(batch, bands_per_frame) + (2 + CH_SHARDS + 8 * groups) * CH_LINE;
}
__global__ __launch_bounds__(256) void k_bilinear_chain(const ChainPlan C, uint8_t *__restrict__ pyramids, size_t stride,
                                                        uint32_t *__restrict__ ctr, uint32_t *__restrict__ hflag,
                                                        uint32_t test) {
  __shared__ uint32_t sh_ok;
  int l = 1;
  while (l + 1 < C.nlevels && (int)blockIdx.x >= C.wg0[l + 1]) l++;
  const int b = (int)blockIdx.x - C.wg0[l];
  const int rest = b >> 3, g = rest / C.wpf[l], chunk = rest - g * C.wpf[l];
  const int frame = 8 * g + (b & 7);
  const int tid = threadIdx.x;
  const int nq = C.nq[l], oh = C.oh[l], nitems = nq * oh;
  const int t0 = chunk * 256, t1 = min(t0 + 256, nitems) - 1;       // this workgroup's items [t0, t1]
  uint32_t *tail = ctr + chain_tail_ofs(C.batch, C.bands_per_frame);
  uint32_t *fault = tail, *top = tail + CH_LINE, *shard = tail + (2 + ((int)blockIdx.x & (CH_SHARDS - 1))) * CH_LINE;
  uint32_t *home = tail + (2 + CH_SHARDS) * CH_LINE;
  const int N = C.kind[l] == 1 ? 8 : 16, M = C.kind[l] == 1 ? 7 : 13;
  const bool idle = frame >= C.batch;               // (the batch is padded to whole groups of eight frames)
  uint32_t *fctr = ctr + (size_t)(idle ? 0 : frame) * C.bands_per_frame;
  if (tid == 0) sh_ok = 1;
  __syncthreads();
  if (!idle) {
    if (tid == 32 && !(test & 32u)) {
      // one XCD per frame (see above): HW_REG_XCC_ID (id 20), bits 3:0
      const uint32_t me = (__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u) + 1u + ((test & 2u) && blockIdx.x == 9 ? 1u : 0u);
      uint32_t seen = __hip_atomic_load(&home[frame * CH_LINE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (seen == 0) {
        uint32_t expect = 0;
        if (!__hip_atomic_compare_exchange_strong(&home[frame * CH_LINE], &expect, me, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
          seen = expect;
        else
          seen = me;
      }
      if (seen != me) {
        __hip_atomic_store(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(hflag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        sh_ok = 0;
      }
    }
    if (l > 1) {
      // source rows of the items' output rows oy0 .. oy1 (level l - 1): block row * N + sy .. + 1; rows the producer never
      // writes (block padding / zero margins below its last written row) are nobody's output
      const int oy0 = t0 / nq, oy1 = t1 / nq;
      const int y0 = oy0 % M, y1 = oy1 % M;
      const int r_lo = (oy0 / M) * N + (M == 7 ? y0 : y0 + (y0 > 3) + (y0 > 8));
      const int r_hi = min((oy1 / M) * N + (M == 7 ? y1 : y1 + (y1 > 3) + (y1 > 8)) + 1, C.oh[l - 1] - 1);
      const int b_lo = r_lo / CH_BAND, b_hi = r_hi / CH_BAND;
      if (tid <= b_hi - b_lo && r_lo <= r_hi && !(test & 4u)) {
        const int band = b_lo + tid;
        const uint32_t want = (uint32_t)(min(CH_BAND, C.oh[l - 1] - band * CH_BAND) * C.nq[l - 1]);
        const uint32_t *p = fctr + C.band0[l - 1] + band;
        bool ok = false;
        const bool poisoned = __hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        const int limit = poisoned ? 0 : 1 << (((test >> 8) & 31u) ? ((test >> 8) & 31u) : 20u);   // (2^20 polls ~ 1 s)
        for (int it = 0; it < limit; it++) {
          if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) {
            ok = true;
            break;
          }
          __builtin_amdgcn_s_sleep(8);
        }
        // (no acquire fence: the items' loads bypass this CU's L1 themselves — bilinear4_item<COH> — and are issued
        //  behind this poll's result in program order, behind the barrier below)
        if (!ok) {
          __hip_atomic_store(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(hflag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          sh_ok = 0;
        }
      }
    }
    __syncthreads();
    if (sh_ok == 0) return;                           // (workgroup-uniform; not counted as done: the launch never re-arms)
    uint8_t *img = pyramids + (size_t)frame * stride;
    const uint8_t *src = img + (size_t)C.row0[l - 1] * C.vstep;
    uint8_t *dst = img + (size_t)C.row0[l] * C.vstep;
    if (C.kind[l] == 1) bilinear4_item<8, 7, true>(src, dst, C.vstep, C.vstep, C.sw[l], C.sh[l], t0 + tid);
    else bilinear4_item<16, 13, true>(src, dst, C.vstep, C.vstep, C.sw[l], C.sh[l], t0 + tid);
    // every thread's stores have been acknowledged by the L2 before the barrier; the counter add behind it needs no
    // release fence (which would write back this XCD's whole L2): the consumers read this L2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (tid == 0) {
    if (!idle && l + 1 < C.nlevels && !((test & 1u) && blockIdx.x == 0) && !(test & 8u)) {
      for (int band = (t0 / nq) / CH_BAND; band <= (t1 / nq) / CH_BAND; band++) {
        const int lo = max(t0, band * CH_BAND * nq), hi = min(t1, (band + 1) * CH_BAND * nq - 1);
        (void)__hip_atomic_fetch_add(&fctr[C.band0[l] + band], (uint32_t)(hi - lo + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // done: CH_SHARDS counters (workgroup b -> shard b % CH_SHARDS); the arrival that completes a shard reports it to the
    // top counter, the one that completes the top is the launch's last workgroup
    uint32_t last = 0;
    if (!(test & 16u)) {
      const uint32_t sh_i = blockIdx.x & (CH_SHARDS - 1);
      const uint32_t expect = (gridDim.x - sh_i + CH_SHARDS - 1) / CH_SHARDS;      // workgroups b < gridDim.x with b % CH_SHARDS == sh_i
      const uint32_t done = __hip_atomic_fetch_add(shard, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (done + 1 == expect) {
        const uint32_t nsh = min((uint32_t)CH_SHARDS, gridDim.x);
        last = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == nsh;
      }
    }
    sh_ok = last ? 2u : 1u;                           // 2: the launch's last workgroup — nobody polls any more
  }
  __syncthreads();
  if (sh_ok == 2u) {
    // re-arm, all 256 threads (ONE thread storing the 6 400 counters of a 64-frame build word by word took ~0.2 ms)
    const int nb = (int)chain_tail_ofs(C.batch, C.bands_per_frame);
    for (int i = tid; i < nb; i += 256) __hip_atomic_store(&ctr[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int i = 1 + tid; i < 2 + CH_SHARDS + 8 * C.groups; i += 256)               // (every line but the sticky fault word's)
      __hip_atomic_store(&tail[i * CH_LINE], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
