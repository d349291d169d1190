/*
 * oracle/pislam_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the arithmetic of the PiSlam ORB front-end hot path
 * (fastDetect -> fastScoreHarris -> fastExtract -> orbCompute).  It is the
 * checker the HIP kernels are compared against; only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may call it.  The shipped
 * library (pislam_amd/csrc) never links or loads this file.
 *
 * Every function cites the reference file:line (paths relative to the
 * 0xfaded/pislam tree) whose behaviour it restates.  The reference is ARMv7
 * NEON; NEON instruction semantics used below are from the ARM Architecture
 * Reference Manual (vhsub/vhadd = floor halving, vmull/vmlal wrap to 16 bit,
 * vrecpe = FPRecipEstimate, vcvt.s32.f32 truncates and maps NaN to 0).
 *
 * PINNING STATUS (see DESIGN.md "Oracle"):
 *   - BRIEF sampler (Brief.h): PINNED against the real reference — Brief.h
 *     contains no NEON and is compiled unmodified into oracle/_ref/ by
 *     oracle/Makefile; tests compare this file's table and descriptors with it.
 *   - keypoint codec (Util.h): PINNED against the real header (oracle/_ref/libutil_ref.so).
 *   - fill_spiral (test/TestUtil.cpp, the fixture of GaussianTest / BilinearTest): PINNED
 *     against the real reference — plain C++, compiled where it lies into oracle/_ref/.
 *   - gaussian5x5 / bilinear (scalar reference() of GaussianTest.cpp / BilinearTest.cpp): restated
 *     from files that include the NEON headers and gtest — not buildable here, unpinned.
 *   - FAST / Harris / NMS / centroid / atan2 (Fast.h, Harris.h, Orb.h):
 *     the reference needs <arm_neon.h> and ARM inline asm, which an x86 image
 *     cannot build without stand-in headers, so there is no oracle/_ref build
 *     for them: "parity unpinned by direct execution".  They are cross-checked
 *     against (a) the SHA-256 prefixes of the reference's outputs on
 *     demo/input.png recorded in SURVEY.md §8c, (b) ARM ARM known answers for
 *     vrecpe, (c) an independent textbook FAST-9 / brute-force circle-moment
 *     formulation in tests/.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* Util.h:27-45 — keypoint codec: score<<24 | x<<12 | y                       */
/* ------------------------------------------------------------------------- */
ORC_API uint32_t orc_encode_fast(uint32_t score, uint32_t x, uint32_t y) {
  return (score << 24) | (x << 12) | y;                      /* Util.h:27-29 */
}
ORC_API uint32_t orc_decode_x(uint32_t e) { return (e >> 12) & 0xfff; }  /* Util.h:35-37 */
ORC_API uint32_t orc_decode_y(uint32_t e) { return e & 0xfff; }          /* Util.h:39-41 */
ORC_API uint32_t orc_decode_score(uint32_t e) { return e >> 24; }        /* Util.h:43-45 */

/* ------------------------------------------------------------------------- */
/* Fast.h:54-158 — fastDetect                                                 */
/* ------------------------------------------------------------------------- */

/* NEON vclz.u8 */
static inline unsigned clz8(unsigned v) {
  unsigned n = 0;
  for (unsigned m = 0x80u; m && !(v & m); m >>= 1) n++;
  return n;
}
/* NEON vshl.u8 by register: signed per-lane count, negative = right shift,
 * |count| >= 8 gives 0 (Fast.h:139,143 use `t << (cnt - 1)` on uint8x16_t,
 * which GCC lowers to vshl.u8 with cnt-1 wrapping to 0xff = -1 when cnt==0). */
static inline unsigned shl8(unsigned v, unsigned count_u8) {
  int s = (int8_t)(uint8_t)count_u8;
  if (s >= 8 || s <= -8) return 0;
  return s >= 0 ? ((v << s) & 0xffu) : ((v & 0xffu) >> (-s));
}

/* One pixel of the segment test; bit layout of d0/l0/d1/l1 as in Fast.h:66-128:
 * ring position k (clockwise from (-3,-1)) -> bit 7-k of d0 for k<8, bit 15-k
 * of d1 for k>=8.  A set bit means "NOT darker"/"NOT brighter". */
static const int8_t RING_DY[16] = {-3,-3,-3,-2,-1, 0, 1, 2, 3, 3, 3, 2, 1, 0,-1,-2};
static const int8_t RING_DX[16] = {-1, 0, 1, 2, 3, 3, 3, 2, 1, 0,-1,-2,-3,-3,-3,-2};

static inline uint8_t fast9_pixel(const uint8_t *img, ptrdiff_t vstep,
                                  ptrdiff_t y, ptrdiff_t x, unsigned thr_u8) {
  unsigned c = img[y * vstep + x];
  unsigned light = c + thr_u8 > 255 ? 255 : c + thr_u8;   /* vqaddq_u8, Fast.h:63 */
  unsigned dark = c < thr_u8 ? 0 : c - thr_u8;             /* vqsubq_u8, Fast.h:64 */
  unsigned d0 = 0, l0 = 0, d1 = 0, l1 = 0;
  for (int k = 0; k < 8; k++) {
    unsigned p = img[(y + RING_DY[k]) * vstep + x + RING_DX[k]];
    unsigned q = img[(y + RING_DY[k + 8]) * vstep + x + RING_DX[k + 8]];
    d0 |= (unsigned)(p >= dark) << (7 - k);                /* vcgeq_u8 */
    l0 |= (unsigned)(p <= light) << (7 - k);               /* vcleq_u8 */
    d1 |= (unsigned)(q >= dark) << (7 - k);
    l1 |= (unsigned)(q <= light) << (7 - k);
  }
  /* Fast.h:130-147 */
  unsigned use_light = (d0 & d1) != 0;                     /* vtstq_u8(d0,d1) */
  unsigned t0 = use_light ? l0 : d0;
  unsigned t1 = use_light ? l1 : d1;
  unsigned cntLo = clz8(t0);
  unsigned testLo = shl8(t1, (cntLo - 1) & 0xffu) == 0 ? 0xffu : 0u;
  unsigned cntHi = clz8(t1);
  unsigned testHi = shl8(t0, (cntHi - 1) & 0xffu) == 0 ? 0xffu : 0u;
  unsigned result = (cntLo & testLo) | (cntHi & testHi);
  return result ? 0xff : 0x00;
}

/* Fast.h:54-158.  Rows [border,height-border); columns in steps of 16 from
 * `border` while x < width-border (Fast.h:60-61), every 16-pixel group fully
 * classified and stored (Fast.h:149), then two zero bytes at out[y][width..]
 * when width%16 != 0 (Fast.h:153-156).  Addressing is flat, as in C. */
ORC_API void orc_fast_detect(int vstep, int border, int width, int height,
                             const uint8_t *img, uint8_t *out, int threshold) {
  unsigned thr = (unsigned)threshold & 0xffu;              /* vdupq_n_u8, Fast.h:58 */
  for (int y = border; y < height - border; y++) {
    for (int x = border; x < width - border; x += 16)
      for (int i = 0; i < 16; i++)
        out[(ptrdiff_t)y * vstep + x + i] = fast9_pixel(img, vstep, y, x + i, thr);
    if (width % 16 != 0) {
      out[(ptrdiff_t)y * vstep + width] = 0;
      out[(ptrdiff_t)y * vstep + width + 1] = 0;
    }
  }
}

/* ------------------------------------------------------------------------- */
/* Harris.h:37-69 harrisEval, Harris.h:80-248 harrisScoreSobel                */
/* ------------------------------------------------------------------------- */
static inline int fl2(int v) { return v >> 1; }            /* floor(v/2): vhsub/vhadd */

ORC_API uint8_t orc_harris_eval(uint32_t Ixx, uint32_t Iyy, int32_t Ixy,
                                int32_t threshold) {
  uint32_t tr = Ixx + Iyy;                                 /* Harris.h:41 */
  tr = tr * tr;                                            /* Harris.h:42 (mod 2^32) */
  tr >>= 4;                                                /* Harris.h:43 */
  uint32_t det = Ixx * Iyy;                                /* Harris.h:46 */
  det = det - (uint32_t)Ixy * (uint32_t)Ixy;               /* vmls_s32, Harris.h:49-50 */
  int32_t score = (int32_t)(det - tr);                     /* Harris.h:53-55 */
  if (threshold < score) {                                 /* Harris.h:58 */
    float f = (float)score;                                /* vcvt_f32_s32, RNE */
    uint32_t bits;
    memcpy(&bits, &f, 4);
    return (uint8_t)((bits >> 20) & 0xff);                 /* Harris.h:63-65 */
  }
  return 0;
}

ORC_API uint8_t orc_harris_score_sobel(int vstep, const uint8_t *img, int x, int y,
                                       int32_t threshold) {
  int P[8][8];
  for (int r = 0; r < 8; r++)                              /* Harris.h:102-110 */
    for (int c = 0; c < 8; c++)
      P[r][c] = img[(ptrdiff_t)(y - 3 + r) * vstep + (x - 3 + c)];

  /* dy: Harris.h:123-135.  D = floor((row[n+2]-row[n])/2) per byte lane, then
   * the [1 2 1]/4 smoothing is two floor-halving adds on lanes j, j+1, j+2. */
  int dy[6][6], dx[6][6];
  for (int n = 0; n < 6; n++) {
    int D[8];
    for (int j = 0; j < 8; j++) D[j] = fl2(P[n + 2][j] - P[n][j]);
    for (int i = 0; i < 6; i++) dy[n][i] = fl2(D[i + 1] + fl2(D[i] + D[i + 2]));
  }
  /* dx: Harris.h:139-162.  E = floor((row[j+2]-row[j])/2), then rows n, n+2
   * are halving-added, then row n+1 (each step uses the RAW E of the two rows
   * below because the macro sequence only ever overwrites row n0). */
  int E[8][6];
  for (int n = 0; n < 8; n++)
    for (int i = 0; i < 6; i++) E[n][i] = fl2(P[n][i + 2] - P[n][i]);
  for (int n = 0; n < 6; n++)
    for (int i = 0; i < 6; i++) dx[n][i] = fl2(fl2(E[n][i] + E[n + 2][i]) + E[n + 1][i]);

  /* Harris.h:166-213: row pairs accumulate in 16-bit lanes (vmull_s8+vmlal_s8
   * wrap), xx/yy re-read as unsigned, xy as signed, widened pairwise. */
  uint32_t sxx = 0, syy = 0;
  int32_t sxy = 0;
  for (int n = 0; n < 6; n += 2)
    for (int i = 0; i < 6; i++) {
      uint16_t xx = (uint16_t)(dx[n][i] * dx[n][i] + dx[n + 1][i] * dx[n + 1][i]);
      uint16_t yy = (uint16_t)(dy[n][i] * dy[n][i] + dy[n + 1][i] * dy[n + 1][i]);
      int16_t xy = (int16_t)(dx[n][i] * dy[n][i] + dx[n + 1][i] * dy[n + 1][i]);
      sxx += xx;
      syy += yy;
      sxy += xy;
    }
  /* Harris.h:243-245 */
  uint32_t Ixx = sxx >> 4, Iyy = syy >> 4;
  int32_t Ixy = sxy >> 4;                                  /* arithmetic */
  return orc_harris_eval(Ixx, Iyy, Ixy, threshold);
}

/* Fast.h:166-180 */
ORC_API void orc_fast_score_harris(int vstep, int border, int width, int height,
                                   const uint8_t *img, int32_t threshold, uint8_t *out) {
  for (int y = border; y < height - border; y++)
    for (int x = border; x < width - border; x++) {
      if (!out[(ptrdiff_t)y * vstep + x]) continue;
      out[(ptrdiff_t)y * vstep + x] = orc_harris_score_sobel(vstep, img, x, y, threshold);
    }
}

/* ------------------------------------------------------------------------- */
/* Fast.h:196-355 — fastExtract (2x2-block NMS + optional bucket top-k)       */
/* Appends to dst[0..cap); returns the number of keypoints the reference      */
/* would have appended (may exceed cap; only the first cap are stored).       */
/* ------------------------------------------------------------------------- */
ORC_API size_t orc_fast_extract(int vstep, int border, int logBucketSize, int bucketLimit,
                                int width, int height, const uint8_t *out,
                                uint32_t *dst, size_t cap) {
  const int bucketSize = 1 << logBucketSize;
  const int numBuckets = (width - 2 * border - 1) / bucketSize + 1;   /* Fast.h:201 */
  size_t n = 0;
  uint32_t *buckets = NULL;
  int *counts = NULL;
  if (logBucketSize != 0 && numBuckets > 0) {
    buckets = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)numBuckets * (size_t)bucketLimit);
    counts = (int *)calloc((size_t)numBuckets, sizeof(int));
  }
#define S(yy, xx) ((unsigned)out[(ptrdiff_t)(yy) * vstep + (xx)])
#define PUSH(v) do { if (n < cap) dst[n] = (v); n++; } while (0)
  for (int y = border; y < height - border; y += 2) {
    if (logBucketSize != 0 && ((y - border) % bucketSize) == 0 && y != border) {
      for (int b = 0; b < numBuckets; b++) {                 /* Fast.h:217-225 */
        for (int i = 0; i < counts[b]; i++) PUSH(buckets[(size_t)b * bucketLimit + i]);
        counts[b] = 0;
      }
    }
    for (int x = border; x < width - border; x += 2) {
      unsigned v0 = S(y, x), v1 = S(y, x + 1), v2 = S(y + 1, x), v3 = S(y + 1, x + 1);
      if (!(v0 | v1 | v2 | v3)) continue;                    /* Fast.h:237 */
      uint32_t result;
      if (v0 > v1 && v0 > v2 && v0 > v3) {                   /* Fast.h:264-273 */
        if (v0 >= S(y - 1, x - 1) && v0 >= S(y, x - 1) && v0 > S(y + 1, x - 1) &&
            v0 >= S(y - 1, x) && v0 >= S(y - 1, x + 1))
          result = orc_encode_fast(v0, (uint32_t)x, (uint32_t)y);
        else
          continue;
      } else if (v1 > v2 && v1 > v3) {                       /* Fast.h:275-285 */
        if (v1 >= S(y - 1, x) && v1 >= S(y - 1, x + 1) && v1 >= S(y - 1, x + 2) &&
            v1 > S(y, x + 2) && v1 > S(y + 1, x + 2))
          result = orc_encode_fast(v1, (uint32_t)x + 1, (uint32_t)y);
        else
          continue;
      } else if (v2 > v3) {                                  /* Fast.h:287-296 */
        if (v2 >= S(y, x - 1) && v2 >= S(y + 1, x - 1) && v2 > S(y + 2, x - 1) &&
            v2 > S(y + 2, x) && v2 > S(y + 2, x + 1))
          result = orc_encode_fast(v2, (uint32_t)x, (uint32_t)y + 1);
        else
          continue;
      } else {                                               /* Fast.h:298-309 */
        if (v3 > S(y + 2, x) && v3 > S(y + 2, x + 1) && v3 >= S(y, x + 2) &&
            v3 > S(y + 1, x + 2) && v3 > S(y + 2, x + 2))
          result = orc_encode_fast(v3, (uint32_t)x + 1, (uint32_t)y + 1);
        else
          continue;
      }
      if (logBucketSize == 0) {                              /* Fast.h:319-320 */
        PUSH(result);
        continue;
      }
      int bucket = (x - border) / bucketSize;                /* Fast.h:316 */
      uint32_t *bk = buckets + (size_t)bucket * bucketLimit;
      int count = counts[bucket];
      if (count < bucketLimit) {                             /* Fast.h:321-333 */
        int i;
        for (i = count - 1; i >= 0 && result < bk[i]; i--) bk[i + 1] = bk[i];
        bk[i + 1] = result;
        counts[bucket] = count + 1;
      } else if (result > bk[0]) {                           /* Fast.h:334-341 */
        int i;
        for (i = 1; i < bucketLimit && result > bk[i]; i++) bk[i - 1] = bk[i];
        bk[i - 1] = result;
      }
    }
  }
  if (logBucketSize != 0)                                    /* Fast.h:345-352 */
    for (int b = 0; b < numBuckets; b++)
      for (int i = 0; i < counts[b]; i++) PUSH(buckets[(size_t)b * bucketLimit + i]);
#undef S
#undef PUSH
  free(buckets);
  free(counts);
  return n;
}

/* ------------------------------------------------------------------------- */
/* Orb.h:80-308 — orbCentroids                                                */
/* ------------------------------------------------------------------------- */
/* Half-width of the patch per |dy|, derived from the masks at Orb.h:118-121
 * and the per-row macros at Orb.h:163-178,208-220,238-250,271-286. */
static const int UMAX[16] = {15,15,15,15,15,15,14,14,13,13,12,11,10,9,7,5};

ORC_API void orc_centroid(int vstep, const uint8_t *img, int x, int y,
                          int32_t *m10, int32_t *m01) {
  int32_t sx = 0, sy = 0;
  for (int dy = -15; dy <= 15; dy++) {
    int u = UMAX[dy < 0 ? -dy : dy];
    const uint8_t *row = img + (ptrdiff_t)(y + dy) * vstep + x;
    for (int dx = -u; dx <= u; dx++) {
      int v = row[dx];
      sx += dx * v;
      sy += dy * v;
    }
  }
  *m10 = sx;
  *m01 = sy;
}

/* Output layout of Orb.h:113-114,298-304: groups of 8 int32 =
 * [x0 x1 x2 x3 y0 y1 y2 y3]; size = roundup8(2*n); padding slots are zero. */
ORC_API size_t orc_centroids_size(size_t n) { return (2 * n + 7) & ~(size_t)7; }

ORC_API void orc_orb_centroids(int vstep, const uint8_t *img, const uint32_t *points,
                               size_t n, int32_t *centroids) {
  memset(centroids, 0, sizeof(int32_t) * orc_centroids_size(n));
  size_t out = 0;
  for (size_t i = 0; i < n; i++) {
    int x = (int)orc_decode_x(points[i]), y = (int)orc_decode_y(points[i]);
    orc_centroid(vstep, img, x, y, &centroids[out], &centroids[out + 4]);
    out += 1;
    if (out % 4 == 0) out += 4;                              /* Orb.h:303-306 */
  }
}

/* ------------------------------------------------------------------------- */
/* Orb.h:310-387 — atan2 -> 30 bins                                           */
/* ------------------------------------------------------------------------- */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* NEON vrecpe.f32 (Orb.h:329) = ARM ARM FPRecipEstimate, 8-bit estimate,
 * flush-to-zero.  Integer form of the pseudocode: a in [0.5,1) has
 * q = floor(512a) = 256 + (frac>>15); r = 1/((q+0.5)/512);
 * s = floor(256 r + 0.5) = (2^19 + (2q+1)) / (2(2q+1)). */
ORC_API float orc_vrecpe(float f) {
  uint32_t u = f2u(f), sign = u & 0x80000000u, e = (u >> 23) & 0xff, m = u & 0x7fffffu;
  if (e == 0xff) return m ? u2f(0x7fc00000u) : u2f(sign);    /* NaN -> default NaN; inf -> 0 */
  if (e == 0) return u2f(sign | 0x7f800000u);                /* 0 / denormal (FTZ) -> inf */
  if (e >= 253) return u2f(sign);                            /* |x| >= 2^126 -> 0 */
  uint32_t q = 256 + (m >> 15);
  uint32_t s = ((1u << 19) + (2 * q + 1)) / (2 * (2 * q + 1));   /* 256..511 */
  return u2f(sign | ((253 - e) << 23) | ((s - 256) << 15));
}

/* NEON vcvt.s32.f32: truncate, saturate, NaN -> 0 (Orb.h:355) */
static inline int32_t cvt_s32_f32(float f) {
  if (f != f) return 0;
  if (f >= 2147483648.0f) return INT32_MAX;
  if (f <= -2147483648.0f) return INT32_MIN;
  return (int32_t)f;
}
/* NEON vmax/vmin.f32 propagate NaN */
static inline float nmax(float a, float b) { return (a != a || b != b) ? NAN : (a > b ? a : b); }
static inline float nmin(float a, float b) { return (a != a || b != b) ? NAN : (a < b ? a : b); }

ORC_API uint8_t orc_angle_bin(int32_t x, int32_t y) {
  /* compile with -ffp-contract=off: ARMv7 NEON has no fused multiply-add here */
  volatile float xf = fabsf((float)x), yf = fabsf((float)y);   /* Orb.h:318-322 */
  float zmax = nmax(xf, yf), zmin = nmin(xf, yf);              /* Orb.h:324-325 */
  volatile float z = zmin * orc_vrecpe(zmax);                  /* Orb.h:327-329 */
  const float c0 = (float)(256 * 14.999998);                   /* Orb.h:336 */
  const float c1 = (float)(256 * 4.723436);                    /* Orb.h:343 */
  const float c2 = (float)(256 * 1.266240);                    /* Orb.h:344 */
  volatile float t0 = c2 * z;
  volatile float t1 = c1 + t0;
  volatile float t2 = z - 1.0f;
  volatile float t3 = t2 * t1;
  volatile float t4 = c0 - t3;
  volatile float af = z * t4;                                  /* Orb.h:345 */
  int32_t angle = cvt_s32_f32(af);                             /* Orb.h:348 */
  int64_t ax = x < 0 ? -(int64_t)x : x, ay = y < 0 ? -(int64_t)y : y;
  if (ax > ay) {                                               /* Orb.h:355-364 */
    if ((x ^ y) < 0) angle = -angle;
    if (x < 0) angle += 256 * 60;
    else if (angle < 0) angle += 256 * 120;
  } else {                                                     /* Orb.h:365-374 */
    if ((x ^ y) >= 0) angle = -angle;
    if (y >= 0) angle += 256 * 30;
    else angle += 256 * 90;
  }
  angle >>= 10;                                                /* Orb.h:376 */
  if (!(0 <= angle && angle < 30)) angle = 0;                  /* Orb.h:377-380 */
  return (uint8_t)angle;
}

/* Orb.h:310-387: consumes groups of 8, emits 4 angles per group (padding
 * slots included), i.e. n8/2 bytes for n8 int32 inputs. */
ORC_API void orc_atan2(const int32_t *xys, size_t n8, uint8_t *angles) {
  for (size_t g = 0; g + 8 <= n8; g += 8)
    for (int i = 0; i < 4; i++) angles[g / 2 + i] = orc_angle_bin(xys[g + i], xys[g + 4 + i]);
}

/* ------------------------------------------------------------------------- */
/* Brief.h:28-53,57-633,637-733 — rotated BRIEF                               */
/* ------------------------------------------------------------------------- */
/* The 256 unrotated test pairs (dx0,dy0,dx1,dy1): OpenCV ORB's bit_pattern_31_
 * (Brief.h:63-67 says so).  The data file is generated by
 * oracle/gen_brief_pattern.py, which recovers the pairs from the COMPILED
 * reference (oracle/_ref) by single-pixel probing — no source text is copied. */
static const int8_t BRIEF_BASE[256][4] = {
#include "brief_pattern_base.inc"
};

static int8_t g_brief_tab[30][256][4];
static int g_brief_ready = 0;

static inline int clamp15(int v) { return v < -15 ? -15 : (v > 15 ? 15 : v); }

/* Brief.h:30-50: theta = float(rot*pi/15) (double product, then float);
 * c,s = cosf/sinf folded by GCC at compile time (correctly rounded float);
 * r = roundf(c*dx - s*dy) etc. with float32 products and sum (dx,dy promoted
 * to float), round half away from zero, clamp to [-15,15]. */
static void brief_build(void) {
  for (int rot = 0; rot < 30; rot++) {
    float theta = (float)(rot * M_PI / 15);
    /* correctly rounded float cos/sin of the float theta: evaluate in long
     * double and round once (verified against the compiled reference's table
     * in tests/test_oracle_brief_ref.py). */
    float c = (float)cosl((long double)theta);
    float s = (float)sinl((long double)theta);
    for (int k = 0; k < 256; k++) {
      const int8_t *b = BRIEF_BASE[k];
      for (int p = 0; p < 2; p++) {
        float dx = (float)b[2 * p], dy = (float)b[2 * p + 1];
        volatile float cx = c * dx, sy = s * dy, sx = s * dx, cy = c * dy;
        volatile float rx = cx - sy, ry = sx + cy;
        g_brief_tab[rot][k][2 * p] = (int8_t)clamp15((int)roundf(rx));
        g_brief_tab[rot][k][2 * p + 1] = (int8_t)clamp15((int)roundf(ry));
      }
    }
  }
  g_brief_ready = 1;
}

/* int8 [30][256][4] = (cdx0, cdy0, cdx1, cdy1) */
ORC_API const int8_t *orc_brief_table(void) {
  if (!g_brief_ready) brief_build();
  return &g_brief_tab[0][0][0];
}

/* Brief.h:637-733 (dispatch) + Brief.h:57-633 (bit k%32 of word k/32) +
 * Brief.h:52 (bit = I(p0) < I(p1)).  rot outside 0..29 writes nothing. */
ORC_API void orc_brief_describe(int vstep, const uint8_t *img, int x, int y, int rot,
                                int words, uint32_t *descriptor) {
  if (rot < 0 || rot >= 30) return;
  const int8_t *tab = orc_brief_table() + (size_t)rot * 256 * 4;
  const uint8_t *base = img + (ptrdiff_t)y * vstep + x;
  for (int w = 0; w < words; w++) {
    uint32_t bits = 0;
    for (int k = 0; k < 32; k++) {
      const int8_t *t = tab + (size_t)(w * 32 + k) * 4;
      unsigned a = base[(ptrdiff_t)t[1] * vstep + t[0]];
      unsigned b = base[(ptrdiff_t)t[3] * vstep + t[2]];
      if (a < b) bits |= 1u << k;
    }
    descriptor[w] = bits;
  }
}

/* ------------------------------------------------------------------------- */
/* Orb.h:396-441 — orbCompute: descriptors[i*words + j] for point i           */
/* ------------------------------------------------------------------------- */
ORC_API void orc_orb_compute(int vstep, int words, const uint8_t *img,
                             const uint32_t *points, size_t n, uint32_t *descriptors) {
  size_t n8 = orc_centroids_size(n);
  int32_t *cen = (int32_t *)malloc(sizeof(int32_t) * (n8 ? n8 : 8));
  uint8_t *ang = (uint8_t *)malloc(n8 / 2 + 4);
  orc_orb_centroids(vstep, img, points, n, cen);
  orc_atan2(cen, n8, ang);
  memset(descriptors, 0, sizeof(uint32_t) * n * (size_t)words);
  for (size_t i = 0; i < n; i++)
    orc_brief_describe(vstep, img, (int)orc_decode_x(points[i]), (int)orc_decode_y(points[i]),
                       ang[i], words, descriptors + i * (size_t)words);
  free(cen);
  free(ang);
}

/* ------------------------------------------------------------------------- */
/* Whole stacked pyramid, the call sequence of demo/demo.cpp:77-101 and        */
/* README.md:67-82: per level detect -> score -> extract (y += level row),     */
/* then one orbCompute over the stacked image.  `score` (rows*vstep) must be   */
/* zero-initialised by the caller (README.md:38, Fast.h:42-44).                */
/* levels = nlevels x {width, height, row0}.  Returns total keypoint count     */
/* (stores at most cap keypoints / descriptors).  level_counts may be NULL.    */
/* ------------------------------------------------------------------------- */
ORC_API size_t orc_pyramid(int vstep, int border, int fast_threshold, int32_t harris_threshold,
                           int logBucketSize, int bucketLimit, int words,
                           const uint8_t *img, uint8_t *score,
                           const int32_t *levels, int nlevels,
                           uint32_t *kp, uint32_t *desc, size_t cap, uint32_t *level_counts) {
  size_t n = 0;
  for (int l = 0; l < nlevels; l++) {
    int w = levels[3 * l], h = levels[3 * l + 1], row0 = levels[3 * l + 2];
    const uint8_t *li = img + (ptrdiff_t)row0 * vstep;
    uint8_t *lo = score + (ptrdiff_t)row0 * vstep;
    orc_fast_detect(vstep, border, w, h, li, lo, fast_threshold);
    orc_fast_score_harris(vstep, border, w, h, li, harris_threshold, lo);
    size_t room = n < cap ? cap - n : 0;
    size_t got = orc_fast_extract(vstep, border, logBucketSize, bucketLimit, w, h, lo,
                                  kp + (n < cap ? n : cap), room);
    size_t stored = got < room ? got : room;
    for (size_t i = 0; i < stored; i++) kp[n + i] += (uint32_t)row0;   /* README.md:78 */
    if (level_counts) level_counts[l] = (uint32_t)got;
    n += got;
  }
  size_t stored = n < cap ? n : cap;
  orc_orb_compute(vstep, words, img, kp, stored, desc);
  return n;
}

/* The same call sequence for levels placed anywhere in the buffer (packed layouts: levels side by side,
 * BASELINE config 4): levels4 = (width, height, row0, col0) per level; README.md:78 adds the level's
 * origin to the keypoint coordinates (x += col0, y += row0). */
ORC_API size_t orc_pyramid4(int vstep, int border, int fast_threshold, int32_t harris_threshold,
                            int logBucketSize, int bucketLimit, int words,
                            const uint8_t *img, uint8_t *score,
                            const int32_t *levels4, int nlevels,
                            uint32_t *kp, uint32_t *desc, size_t cap, uint32_t *level_counts) {
  size_t n = 0;
  for (int l = 0; l < nlevels; l++) {
    int w = levels4[4 * l], h = levels4[4 * l + 1], row0 = levels4[4 * l + 2], col0 = levels4[4 * l + 3];
    const uint8_t *li = img + (ptrdiff_t)row0 * vstep + col0;
    uint8_t *lo = score + (ptrdiff_t)row0 * vstep + col0;
    orc_fast_detect(vstep, border, w, h, li, lo, fast_threshold);
    orc_fast_score_harris(vstep, border, w, h, li, harris_threshold, lo);
    size_t room = n < cap ? cap - n : 0;
    size_t got = orc_fast_extract(vstep, border, logBucketSize, bucketLimit, w, h, lo,
                                  kp + (n < cap ? n : cap), room);
    size_t stored = got < room ? got : room;
    for (size_t i = 0; i < stored; i++) kp[n + i] += ((uint32_t)col0 << 12) | (uint32_t)row0;
    if (level_counts) level_counts[l] = (uint32_t)got;
    n += got;
  }
  size_t stored = n < cap ? n : cap;
  orc_orb_compute(vstep, words, img, kp, stored, desc);
  return n;
}

/* cpu_baseline leg of bench.py: the same path on `nthreads` host threads, one pyramid
 * per thread at a time, for at least `min_seconds` (every thread keeps taking pyramids, round robin over
 * the `npyr` inputs, until the time is up).  Returns the keypoints produced; *pyramids_done and
 * *seconds report the sample.  Each thread owns its score map and output buffers. */
#include <pthread.h>
#include <time.h>
typedef struct {
  int vstep, border, fast_threshold, logBucketSize, bucketLimit, words, nlevels, npyr, tid, nthreads;
  int32_t harris_threshold;
  const uint8_t *imgs;
  size_t img_stride, rows, cap;
  const int32_t *levels4;
  double deadline, t_end;
  unsigned long long kps, pyrs;
} orc_mt_job;
static double orc_now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static void *orc_mt_worker(void *arg) {
  orc_mt_job *j = (orc_mt_job *)arg;
  uint8_t *score = (uint8_t *)calloc(j->rows * (size_t)j->vstep, 1);
  uint32_t *kp = (uint32_t *)malloc(sizeof(uint32_t) * j->cap);
  uint32_t *desc = (uint32_t *)malloc(sizeof(uint32_t) * j->cap * (size_t)j->words);
  int i = j->tid % j->npyr;
  do {
    memset(score, 0, j->rows * (size_t)j->vstep);          /* Fast.h:42-44: `out` starts as zeros */
    j->kps += orc_pyramid4(j->vstep, j->border, j->fast_threshold, j->harris_threshold, j->logBucketSize,
                           j->bucketLimit, j->words, j->imgs + (size_t)i * j->img_stride, score, j->levels4,
                           j->nlevels, kp, desc, j->cap, NULL);
    j->pyrs++;
    i = (i + j->nthreads) % j->npyr;
  } while (orc_now() < j->deadline);
  j->t_end = orc_now();
  free(score);
  free(kp);
  free(desc);
  return NULL;
}
ORC_API unsigned long long orc_pyramid_mt(int nthreads, double min_seconds, int vstep, int rows, int border,
                                          int fast_threshold, int32_t harris_threshold, int logBucketSize,
                                          int bucketLimit, int words, const uint8_t *imgs, size_t img_stride,
                                          int npyr, const int32_t *levels4, int nlevels, size_t cap,
                                          unsigned long long *pyramids_done, double *seconds) {
  if (nthreads < 1) nthreads = 1;
  orc_mt_job *jobs = (orc_mt_job *)calloc((size_t)nthreads, sizeof(orc_mt_job));
  pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
  const double t0 = orc_now();
  for (int t = 0; t < nthreads; t++) {
    orc_mt_job *j = &jobs[t];
    j->vstep = vstep; j->border = border; j->fast_threshold = fast_threshold; j->logBucketSize = logBucketSize;
    j->bucketLimit = bucketLimit; j->words = words; j->nlevels = nlevels; j->npyr = npyr; j->tid = t;
    j->nthreads = nthreads; j->harris_threshold = harris_threshold; j->imgs = imgs; j->img_stride = img_stride;
    j->rows = (size_t)rows; j->cap = cap; j->levels4 = levels4; j->deadline = t0 + min_seconds;
    if (pthread_create(&th[t], NULL, orc_mt_worker, j) != 0) { nthreads = t; break; }
  }
  unsigned long long kps = 0, pyrs = 0;
  double t_end = t0;
  for (int t = 0; t < nthreads; t++) {
    pthread_join(th[t], NULL);
    kps += jobs[t].kps;
    pyrs += jobs[t].pyrs;
    if (jobs[t].t_end > t_end) t_end = jobs[t].t_end;
  }
  if (pyramids_done) *pyramids_done = pyrs;
  if (seconds) *seconds = t_end - t0;
  free(jobs);
  free(th);
  return kps;
}

/* ========================================================================= */
/* "Next" tier: image preparation.  The reference's NEON/asm  */
/* (Gaussian.h, Bilinear.h) cannot be built here; its OWN tests state the       */
/* expected arithmetic as scalar reference functions, restated below.           */
/* ========================================================================= */

/* test/GaussianTest.cpp:32 — RHADD(a,b) = (a>>1)+(b>>1)+((a|b)&1) = (a+b+1)>>1 */
static inline unsigned rhadd(unsigned a, unsigned b) { return (a >> 1) + (b >> 1) + ((a | b) & 1u); }

/* test/GaussianTest.cpp:159-215 `reference()`: in-place separable [1 4 6 4 1]/16 built from RHADDs,
 * vertical pass then horizontal pass on its result, reflect-101 borders.  (pislam::gaussian5x5<vstep>,
 * Gaussian.h:48, is asserted equal to this on the width x height region.) */
ORC_API void orc_gaussian5x5(int vstep, int width, int height, uint8_t *m) {
  for (int j = 0; j < width; j++) {                          /* GaussianTest.cpp:162-186 */
    unsigned a = m[2 * vstep + j], b = m[1 * vstep + j], c = m[j], d = m[1 * vstep + j], e;
    for (int i = 0; i < height; i++) {
      if (i == height - 2) e = c;
      else if (i == height - 1) e = a;
      else e = m[(ptrdiff_t)(i + 2) * vstep + j];
      unsigned x = rhadd(a, e), y = rhadd(b, d);
      x = rhadd(x, c);
      x = rhadd(x, c);
      m[(ptrdiff_t)i * vstep + j] = (uint8_t)rhadd(x, y);
      a = b; b = c; c = d; d = e;
    }
  }
  for (int i = 0; i < height; i++) {                         /* GaussianTest.cpp:189-213 */
    uint8_t *r = m + (ptrdiff_t)i * vstep;
    unsigned a = r[2], b = r[1], c = r[0], d = r[1], e;
    for (int j = 0; j < width; j++) {
      if (j == width - 2) e = c;
      else if (j == width - 1) e = a;
      else e = r[j + 2];
      unsigned x = rhadd(a, e), y = rhadd(b, d);
      x = rhadd(x, c);
      x = rhadd(x, c);
      r[j] = (uint8_t)rhadd(x, y);
      a = b; b = c; c = d; d = e;
    }
  }
}

/* test/BilinearTest.cpp:35 — RSHR(a,n) = (a>>n) + ((a>>(n-1))&1) = (a + 2^(n-1)) >> n */
static inline int rshr8(int a) { return (a >> 8) + ((a >> 7) & 1); }

/* test/BilinearTest.cpp:171-196 `reference7_8()`: in place, per 8x8 block 7x7 outputs.
 * (pislam::bilinear7_8<vstep>, Bilinear.h:42, is asserted equal on the (h*7/8) x (w*7/8) region.) */
ORC_API void orc_bilinear7_8(int vstep, int width, int height, uint8_t *m) {
  static const int f[7] = {238, 201, 165, 128, 91, 55, 18};
  for (int i = 0, oi = 0; i < height; i += 8, oi += 7)
    for (int j = 0, oj = 0; j < width; j += 8, oj += 7)
      for (int y = 0; y < 7; y++)
        for (int x = 0; x < 7; x++) {
          const int p00 = m[(ptrdiff_t)vstep * (i + y) + (j + x)], p01 = m[(ptrdiff_t)vstep * (i + y) + (j + x + 1)];
          const int p10 = m[(ptrdiff_t)vstep * (i + y + 1) + (j + x)], p11 = m[(ptrdiff_t)vstep * (i + y + 1) + (j + x + 1)];
          const int h0 = rshr8(p00 * f[x] + p01 * f[6 - x]);
          const int h1 = rshr8(p10 * f[x] + p11 * f[6 - x]);
          m[(ptrdiff_t)vstep * (oi + y) + (oj + x)] = (uint8_t)rshr8(h0 * f[y] + h1 * f[6 - y]);
        }
}

static inline int map13(int i) {                             /* BilinearTest.cpp:198-206 */
  if (i > 3) i += 1;
  if (i > 9) i += 1;
  return i;
}

/* test/BilinearTest.cpp:208-233 `reference13_16()` (note f[10] = 138, as in Bilinear.h:172-180). */
ORC_API void orc_bilinear13_16(int vstep, int width, int height, uint8_t *m) {
  static const int f[13] = {226, 167, 108, 49, 246, 187, 128, 69, 10, 207, 138, 89, 30};
  for (int i = 0, oi = 0; i < height; i += 16, oi += 13)
    for (int j = 0, oj = 0; j < width; j += 16, oj += 13)
      for (int y = 0; y < 13; y++)
        for (int x = 0; x < 13; x++) {
          const int yy = i + map13(y), xx = j + map13(x);
          const int p00 = m[(ptrdiff_t)vstep * yy + xx], p01 = m[(ptrdiff_t)vstep * yy + xx + 1];
          const int p10 = m[(ptrdiff_t)vstep * (yy + 1) + xx], p11 = m[(ptrdiff_t)vstep * (yy + 1) + xx + 1];
          const int h0 = rshr8(p00 * f[x] + p01 * f[12 - x]);
          const int h1 = rshr8(p10 * f[x] + p11 * f[12 - x]);
          m[(ptrdiff_t)vstep * (oi + y) + (oj + x)] = (uint8_t)rshr8(h0 * f[y] + h1 * f[12 - y]);
        }
}

/* test/TestUtil.cpp:28-55 fill_spiral — the fixture of GaussianTest / BilinearTest (golden-ratio spiral
 * of 0xff on zeros; float32 math as in the reference). */
ORC_API void orc_fill_spiral(int vstep, int width, int height, int cx, int cy, uint8_t *buffer) {
  (void)width;
  memset(buffer, 0, (size_t)vstep * (size_t)height);
  const float phi = (1 + sqrtf(5)) / 2;
  for (float theta = 0; theta < 20; theta += 0.01f) {
    const float r = powf(phi, (float)(theta * M_2_PI));
    const float x = r * cosf(theta), y = r * sinf(theta);
    int i = (int)(y + cy), j = (int)(x + cx);
    if (0 <= i && i < vstep && 0 <= j && j < vstep && i < height) buffer[i * vstep + j] = 0xff;
    i = (int)(-y + cy);
    j = (int)(-x + cx);
    if (0 <= i && i < vstep && 0 <= j && j < vstep && i < height) buffer[i * vstep + j] = 0xff;
  }
}

/* test/TestUtil.cpp:57-65 fill_random — the other fixture of BilinearTest (random7_8 / random13_16,
 * BilinearTest.cpp:85-95,139-149) and GaussianTest: a default-constructed std::mt19937_64 (seed 5489,
 * the C++11 standard's mersenne_twister_engine<uint64, 64,312,156,31, 0xb5026f5aa96619e9, 29,
 * 0x5555555555555555, 17, 0x71d67fffeda60000, 37, 0xfff7eee000000000, 43, 6364136223846793005>),
 * one draw per pixel in raster order, truncated to a byte.  PINNED against the reference's own
 * TestUtil.cpp compiled into oracle/_ref/libtestutil_ref.so (tests/test_oracle.py). */
ORC_API void orc_fill_random(int vstep, int width, int height, uint8_t *buffer) {
  enum { NN = 312, MM = 156 };
  static const uint64_t UM = 0xFFFFFFFF80000000ull, LM = 0x7FFFFFFFull, MATRIX_A = 0xB5026F5AA96619E9ull;
  uint64_t mt[NN];
  mt[0] = 5489ull;
  for (int i = 1; i < NN; i++) mt[i] = 6364136223846793005ull * (mt[i - 1] ^ (mt[i - 1] >> 62)) + (uint64_t)i;
  int mti = NN;
  for (int i = 0; i < height; i++)
    for (int j = 0; j < width; j++) {
      if (mti >= NN) {
        for (int k = 0; k < NN; k++) {
          const uint64_t x = (mt[k] & UM) | (mt[(k + 1) % NN] & LM);
          mt[k] = mt[(k + MM) % NN] ^ (x >> 1) ^ ((x & 1ull) ? MATRIX_A : 0ull);
        }
        mti = 0;
      }
      uint64_t x = mt[mti++];
      x ^= (x >> 29) & 0x5555555555555555ull;
      x ^= (x << 17) & 0x71D67FFFEDA60000ull;
      x ^= (x << 37) & 0xFFF7EEE000000000ull;
      x ^= (x >> 43);
      buffer[(ptrdiff_t)i * vstep + j] = (uint8_t)x;
    }
}

/* Brute-force Hamming matcher.  NOT a restatement of reference code: the
 * reference ships no matcher (README.md:125-128 only names matching as the consumer of the
 * descriptors), so this function DEFINES the semantics that pislam_match_hamming (include/
 * pislam_hip.h) implements on the GPU: per query the train index of minimum Hamming distance (ties
 * -> smallest index; -1 when nt == 0), that distance, and the minimum distance among the other
 * train descriptors (0xffffffff when there is none). */
ORC_API void orc_match_hamming(int words, const uint32_t *query, size_t nq, const uint32_t *train, size_t nt,
                               int32_t *idx, uint32_t *dist, uint32_t *dist2) {
  for (size_t i = 0; i < nq; i++) {
    int32_t bi = -1;
    uint32_t bd = 0xffffffffu, sd = 0xffffffffu;
    for (size_t j = 0; j < nt; j++) {
      uint32_t d = 0;
      for (int k = 0; k < words; k++) d += (uint32_t)__builtin_popcount(query[i * words + k] ^ train[j * words + k]);
      if (d < bd) {
        sd = bd;
        bd = d;
        bi = (int32_t)j;
      } else if (d < sd) {
        sd = d;
      }
    }
    idx[i] = bi;
    dist[i] = bd;
    dist2[i] = sd;
  }
}
