// pislam_dev.h — device-side primitives of the ORB front-end (gfx950, wave64).
//
// Every function states which reference lines define its result; the bodies are
// written for the CDNA4 execution model (per-lane integer ALU, wave ballots,
// LDS tiles), not transliterated from the NEON code.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pdev {

// ---------------------------------------------------------------------------
// Keypoint codec — reference Util.h:27-45
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t encode_fast(uint32_t s, uint32_t x, uint32_t y) {
  return (s << 24) | (x << 12) | y;
}
__device__ __forceinline__ int decode_x(uint32_t e) { return (e >> 12) & 0xfff; }
__device__ __forceinline__ int decode_y(uint32_t e) { return e & 0xfff; }

// ---------------------------------------------------------------------------
// FAST-9 segment test — result of reference Fast.h:63-147 for one pixel.
// Ring positions clockwise from (dy,dx)=(-3,-1) (Fast.h:66-128).  A pixel is
// "dark" iff p < c - t and "bright" iff p > c + t (the saturating forms at
// Fast.h:63-64 are equivalent); corner iff 9 circularly contiguous dark or 9
// contiguous bright positions (what the clz/shift test at Fast.h:130-147
// decides).
// ---------------------------------------------------------------------------
#define PISLAM_RING16(F)                                                         \
  F(0, -3, -1) F(1, -3, 0) F(2, -3, 1) F(3, -2, 2) F(4, -1, 3) F(5, 0, 3)        \
  F(6, 1, 3) F(7, 2, 2) F(8, 3, 1) F(9, 3, 0) F(10, 3, -1) F(11, 2, -2)          \
  F(12, 1, -3) F(13, 0, -3) F(14, -1, -3) F(15, -2, -2)

// 16-bit circular mask -> does it contain a run of >= 9 set bits?
__device__ __forceinline__ bool has_arc9(uint32_t m) {
  uint32_t x = m | (m << 16);
  uint32_t r = x & (x >> 1);   // runs of 2
  r &= r >> 2;                 // runs of 4
  r &= r >> 4;                 // runs of 8
  r &= x >> 8;                 // runs of 9
  return (r & 0xffffu) != 0;
}

// c points at the centre pixel; pitch in bytes; thr already truncated to 8 bits.
template <class P>
__device__ __forceinline__ bool fast9(const P *c, int pitch, int thr) {
  const int v = c[0];
  const int lo = v - thr, hi = v + thr;
  // The sign bit of (p - lo) / (hi - p) IS the dark / bright flag; v_alignbit_b32 shifts it into the
  // mask in one instruction ({mask, diff} >> 31).  Positions enter in ring order, so the 16-bit
  // circular mask is a rotation/reflection-free image of the ring (bit 15-k = position k).
  uint32_t dm = 0, bm = 0;
#define PISLAM_F(k, dy, dx)                                                  \
  {                                                                          \
    const int p = c[(dy) * pitch + (dx)];                                    \
    dm = __builtin_amdgcn_alignbit(dm, (uint32_t)(p - lo), 31);              \
    bm = __builtin_amdgcn_alignbit(bm, (uint32_t)(hi - p), 31);              \
  }
  PISLAM_RING16(PISLAM_F)
#undef PISLAM_F
  return has_arc9(dm) || has_arc9(bm);
}

// The same decision with the multimedia byte instructions: the 16 ring pixels are packed 4 per dword
// and compared 4 at a time with v_lerp_u8 — lerp(a, b, r) = (a + b + r) >> 1 per byte, whose bit 7 is
// [a + b + r >= 256]:
//   bright  p > c + t   <=>  p + max(255 - t - c, 0) >= 256        (r = 0; with c + t >= 255 the addend is 0 and the
//                                                                    sum never reaches 256: no bright pixel exists)
//   NOT dark  p >= lo, lo = max(c - t, 0) (the reference's saturating form, Fast.h:63-64)
//                       <=>  p + (255 - lo) + 1 >= 256              (r = 1), 255 - lo = min(255 + t - c, 255).
// Three instructions per threshold dword (subtract, clamp, replicate), no special case.  The four flag bits of a
// dword are gathered with v_dot4_u32_u8 (weights = bit positions, flags are 0x80 -> the sum is 128 x the
// bits).  ~40 VALU for the two 16-bit masks instead of 64 (one subtract + one v_alignbit per pixel and
// side in fast9); identical result.
template <class P>
__device__ __forceinline__ bool fast9_mm(const P *c, int pitch, int thr) {
  const int v = c[0];
  // (thr is wave-uniform in every caller; the opaque copies keep 255 -/+ thr in scalar registers, one subtract per lane)
  int k_hi = 255 - thr, k_lo = 255 + thr;
  asm("" : "+s"(k_hi), "+s"(k_lo));
  const uint32_t bhi4 = (uint32_t)max(k_hi - v, 0) * 0x01010101u;
  const uint32_t nlo4 = (uint32_t)min(k_lo - v, 255) * 0x01010101u;
  uint32_t px[16];
#define PISLAM_F(k, dy, dx) px[k] = c[(dy) * pitch + (dx)];
  PISLAM_RING16(PISLAM_F)
#undef PISLAM_F
  uint32_t tb[2] = {0u, 0u}, td[2] = {0u, 0u};         // 128 x the low / high 8 mask bits
#pragma unroll
  for (int j = 0; j < 4; j++) {
    // 4 zero-extended bytes -> one dword with three v_perm_b32
    const uint32_t t01 = __builtin_amdgcn_perm(px[4 * j + 1], px[4 * j], 0x0c0c0400u);
    const uint32_t t23 = __builtin_amdgcn_perm(px[4 * j + 3], px[4 * j + 2], 0x0c0c0400u);
    const uint32_t w = __builtin_amdgcn_perm(t23, t01, 0x05040100u);
    const uint32_t gb = __builtin_amdgcn_lerp(w, bhi4, 0u);               // bit 7 of a byte: p > c + t (bright)
    const uint32_t nd = __builtin_amdgcn_lerp(w, nlo4, 0x01010101u);      // bit 7 of a byte: p >= lo (not dark)
    const uint32_t wt = (j & 1) ? 0x80402010u : 0x08040201u;
    tb[j >> 1] = __builtin_amdgcn_udot4(gb & 0x80808080u, wt, tb[j >> 1], false);
    td[j >> 1] = __builtin_amdgcn_udot4(~nd & 0x80808080u, wt, td[j >> 1], false);
  }
  // Both 16-bit ring masks in ONE dword (dark low, bright high) and the 9-run test on both halves at once with
  // packed 16-bit shifts: a rotation inside each half is v_pk_lshrrev_b16 | v_pk_lshlrev_b16, the rotation by 8 a
  // byte swap (v_perm_b32).  Same decision as has_arc9(dm) || has_arc9(bm) — whose `||` hipcc turned into a branch
  // around the second side that is taken by nearly every batch — in 12 instead of 24 VALU.
  typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
  // (td / tb = 128 x the mask bytes: low and high byte joined by one shift-add each, then the two sides by a shift and a
  //  shift-or — four instructions; the four-term OR of shifted sums took six)
  const uint32_t dsum = td[0] + (td[1] << 8), bsum = tb[0] + (tb[1] << 8);
  uint32_t pm = (dsum >> 7) | (bsum << 9);
  auto rotr16x2 = [](uint32_t x, int k) -> uint32_t {
    const us2_t v = __builtin_bit_cast(us2_t, x);
    const us2_t a = v >> (us2_t)((unsigned short)k), b = v << (us2_t)((unsigned short)(16 - k));
    return __builtin_bit_cast(uint32_t, a) | __builtin_bit_cast(uint32_t, b);
  };
  uint32_t r = pm & rotr16x2(pm, 1);                             // runs of 2 (circular, per half)
  r &= rotr16x2(r, 2);                                           // runs of 4
  r &= rotr16x2(r, 4);                                           // runs of 8
  r &= __builtin_amdgcn_perm(0u, pm, 0x02030001u);               // runs of 9: the mask rotated by 8 = bytes swapped per half
  return r != 0;
}

// ---------------------------------------------------------------------------
// Harris 6x6 Sobel score byte — reference Harris.h:80-248 + harrisEval
// Harris.h:37-69.  c points at img[y][x].
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint8_t harris_eval(uint32_t Ixx, uint32_t Iyy, int32_t Ixy,
                                               int32_t threshold) {
  uint32_t tr = Ixx + Iyy;                                 // Harris.h:41
  tr = (tr * tr) >> 4;                                     // Harris.h:42-43
  uint32_t det = Ixx * Iyy - (uint32_t)Ixy * (uint32_t)Ixy;   // Harris.h:46-50
  int32_t score = (int32_t)(det - tr);                     // Harris.h:53-55
  if (threshold < score) {                                 // Harris.h:58
    float f = (float)score;                                // v_cvt_f32_i32: RNE like vcvt.f32.s32
    return (uint8_t)((__float_as_uint(f) >> 20) & 0xff);   // Harris.h:63-65
  }
  return 0;
}

template <class P>
__device__ __forceinline__ uint8_t harris_score(const P *c, int pitch, int32_t threshold) {
  int px[8][8];
#pragma unroll
  for (int r = 0; r < 8; r++)
#pragma unroll
    for (int q = 0; q < 8; q++) px[r][q] = c[(r - 3) * pitch + (q - 3)];   // Harris.h:102-110
  uint32_t sxx = 0, syy = 0;
  int32_t sxy = 0;
#pragma unroll
  for (int n = 0; n < 6; n += 2) {
#pragma unroll
    for (int i = 0; i < 6; i++) {
      int dxv[2], dyv[2];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int m = n + h;
        // dx: Harris.h:139-162 (halving adds; rows m, m+2 first, then m+1)
        const int e0 = (px[m][i + 2] - px[m][i]) >> 1;
        const int e1 = (px[m + 1][i + 2] - px[m + 1][i]) >> 1;
        const int e2 = (px[m + 2][i + 2] - px[m + 2][i]) >> 1;
        dxv[h] = (((e0 + e2) >> 1) + e1) >> 1;
        // dy: Harris.h:123-135 (lanes i, i+2 first, then i+1)
        const int d0 = (px[m + 2][i] - px[m][i]) >> 1;
        const int d1 = (px[m + 2][i + 1] - px[m][i + 1]) >> 1;
        const int d2 = (px[m + 2][i + 2] - px[m][i + 2]) >> 1;
        dyv[h] = (d1 + ((d0 + d2) >> 1)) >> 1;
      }
      // Harris.h:166-213: 16-bit lane accumulation of a row pair, then widening
      sxx += (uint32_t)(dxv[0] * dxv[0] + dxv[1] * dxv[1]) & 0xffffu;
      syy += (uint32_t)(dyv[0] * dyv[0] + dyv[1] * dyv[1]) & 0xffffu;
      sxy += (int32_t)(int16_t)(dxv[0] * dyv[0] + dxv[1] * dyv[1]);
    }
  }
  return harris_eval(sxx >> 4, syy >> 4, sxy >> 4, threshold);   // Harris.h:243-247
}

// The same score on packed signed 16-bit pairs: one register holds two horizontally adjacent
// columns, so the halving adds, the 16-bit wrapped row-pair products of Harris.h:166-213 (exactly
// v_pk_mul_lo_u16 / v_pk_mad_u16 semantics) and their widening sums (v_dot2) each cover two columns
// per instruction.  Rows are fetched as two (byte-unaligned) dwords.  ~40 % fewer instructions and
// far fewer live registers than the scalar form; results are identical (GPU parity tests).
typedef short pk_s2 __attribute__((ext_vector_type(2)));
typedef unsigned short pk_u2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) uint8_t lds_byte;
typedef __attribute__((address_space(3))) uint32_t lds_word;
// row0 = &tile[y-3][x-3] in LDS, any byte alignment.  gfx950 serves byte-unaligned ds_read_b32 but
// stalls ~47 cycles on each (SQ_LDS_UNALIGNED_STALL), so every row is fetched as three aligned
// dwords and funnel-shifted with v_alignbyte.
//
// Harris 6x6 for one corner with the gradients computed in the BYTE domain, as the reference's NEON code does
// (vhsub.u8 / vhadd.s8 on 8-bit lanes, Harris.h:123-162), 4 pixels per instruction:
//   floor((a - b) / 2) + 128 = v_lerp_u8(a, ~b, 1)      (a - b + 256) >> 1 — never leaves a byte)
//   floor((e + f) / 2) + 128 = v_lerp_u8(E, F, 0)       for E = e + 128, F = f + 128
// so a difference and both halving adds of the Sobel chains are single instructions on offset-binary
// bytes, and the products are v_dot4_i32_i8 (see below why the reference's 16-bit product lanes never
// wrap).  ~250 VALU per 64 corners (a packed-i16 formulation of the same arithmetic took ~430);
// result-identical to the scalar harris_score above.
__device__ __forceinline__ uint8_t harris_score_mm(const lds_byte *row0, int pitch_bytes, int32_t threshold) {
  const uint32_t sh = (uint32_t)(uintptr_t)row0 & 3u;     // tile base and pitch are 16-byte aligned
  const lds_byte *base = row0 - sh;
  constexpr uint32_t ONE4 = 0x01010101u;
  uint32_t w0[8], w1[8], n0[8], n1[8], E0[8], E1[8];
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const lds_word *rp = (const lds_word *)(base + r * pitch_bytes);
    const uint32_t i0 = rp[0], i1 = rp[1], i2 = rp[2];
    w0[r] = __builtin_amdgcn_alignbyte(i1, i0, sh);        // window columns 0..3 of row r
    w1[r] = __builtin_amdgcn_alignbyte(i2, i1, sh);        // columns 4..7
    n0[r] = ~w0[r];
    n1[r] = ~w1[r];
    // horizontal differences (P[c+2] - P[c]) >> 1 of columns c = 0..3 and 4..5 (bytes 2,3 of E1 unused)
    E0[r] = __builtin_amdgcn_lerp(__builtin_amdgcn_alignbyte(w1[r], w0[r], 2), n0[r], ONE4);
    E1[r] = __builtin_amdgcn_lerp(w1[r] >> 16, n1[r], ONE4);
  }
  // Products.  The reference multiplies on 16-bit lanes (vmull.s8 / vmlal.s8 per row pair, then widening
  // adds, Harris.h:164-200), but those lanes can never wrap: a gradient is in [-128, 127], so a row pair
  // of squares is <= 2 * 16384 (fits u16), and dx = dy = -128 at one position is impossible — dx = -128
  // needs P[m..m+2][c] = 255, P[..][c+2] = 0 while dy = -128 needs P[m+2][c..c+2] = 0 — so a row pair of
  // dx * dy stays within [-32512, 32512].  The three sums are therefore plain sums of byte products:
  // v_dot4_i32_i8 on the two's-complement bytes (offset-binary ^ 0x80), 4 positions per instruction.
  int32_t sxx = 0, syy = 0, sxy = 0;
  // Columns 0..3 of a row fill a dword; columns 4, 5 only half of one.  Rows are therefore taken in PAIRS: the
  // column-4/5 gradients of rows m and m + 1 share one dword (three v_perm_b32 pick the byte pairs of the two rows'
  // vertical differences that the dy chain combines, one more joins the two dx results), so that their halving adds,
  // sign conversion and dot products are paid once per pair — 11 instead of 18 instructions per row pair.
#pragma unroll
  for (int m = 0; m < 6; m += 2) {
    uint32_t DX1[2], V1[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int q = m + h;
      // dx: (((e_q + e_q+2) >> 1) + e_q+1) >> 1, Harris.h:139-162
      const uint32_t DX0 = __builtin_amdgcn_lerp(__builtin_amdgcn_lerp(E0[q], E0[q + 2], 0u), E0[q + 1], 0u);
      DX1[h] = __builtin_amdgcn_lerp(__builtin_amdgcn_lerp(E1[q], E1[q + 2], 0u), E1[q + 1], 0u);
      // vertical differences (P[q+2] - P[q]) >> 1 of columns 0..3 and 4..7, then
      // dy: (d_c+1 + ((d_c + d_c+2) >> 1)) >> 1, Harris.h:123-135
      const uint32_t V0 = __builtin_amdgcn_lerp(w0[q + 2], n0[q], ONE4);
      V1[h] = __builtin_amdgcn_lerp(w1[q + 2], n1[q], ONE4);
      const uint32_t DY0 = __builtin_amdgcn_lerp(__builtin_amdgcn_lerp(V0, __builtin_amdgcn_alignbyte(V1[h], V0, 2), 0u),
                                                 __builtin_amdgcn_alignbyte(V1[h], V0, 1), 0u);
      // signed bytes of columns 0..3
      const int x0 = (int)(DX0 ^ 0x80808080u), y0 = (int)(DY0 ^ 0x80808080u);
      sxx = __builtin_amdgcn_sdot4(x0, x0, sxx, false);
      syy = __builtin_amdgcn_sdot4(y0, y0, syy, false);
      sxy = __builtin_amdgcn_sdot4(x0, y0, sxy, false);
    }
    // columns 4, 5 of both rows: bytes {row m: c4, c5, row m+1: c4, c5}
    const uint32_t DX1p = __builtin_amdgcn_perm(DX1[1], DX1[0], 0x05040100u);
    const uint32_t va = __builtin_amdgcn_perm(V1[1], V1[0], 0x05040100u);     // d_c   (columns 4, 5)
    const uint32_t vb = __builtin_amdgcn_perm(V1[1], V1[0], 0x07060302u);     // d_c+2 (columns 6, 7)
    const uint32_t vc = __builtin_amdgcn_perm(V1[1], V1[0], 0x06050201u);     // d_c+1 (columns 5, 6)
    const uint32_t DY1p = __builtin_amdgcn_lerp(__builtin_amdgcn_lerp(va, vb, 0u), vc, 0u);
    const int x1 = (int)(DX1p ^ 0x80808080u), y1 = (int)(DY1p ^ 0x80808080u);
    sxx = __builtin_amdgcn_sdot4(x1, x1, sxx, false);
    syy = __builtin_amdgcn_sdot4(y1, y1, syy, false);
    sxy = __builtin_amdgcn_sdot4(x1, y1, sxy, false);
  }
  return harris_eval((uint32_t)sxx >> 4, (uint32_t)syy >> 4, sxy >> 4, threshold);
}

// ---------------------------------------------------------------------------
// 2x2-block non-max suppression — reference Fast.h:228-312.
// s points at S[y][x] (block origin).  Returns the packed keypoint
// encode(v, x', y') or 0 when the block emits nothing.
// ---------------------------------------------------------------------------
template <class P>
__device__ __forceinline__ uint32_t nms_block(const P *s, int pitch, int x, int y) {
#define S_(dy, dx) ((uint32_t)s[(dy) * pitch + (dx)])
  const uint32_t v0 = S_(0, 0), v1 = S_(0, 1), v2 = S_(1, 0), v3 = S_(1, 1);
  if (!(v0 | v1 | v2 | v3)) return 0;                                   // Fast.h:237
  if (v0 > v1 && v0 > v2 && v0 > v3) {                                  // Fast.h:264-273
    if (v0 >= S_(-1, -1) && v0 >= S_(0, -1) && v0 > S_(1, -1) && v0 >= S_(-1, 0) &&
        v0 >= S_(-1, 1))
      return encode_fast(v0, x, y);
    return 0;
  } else if (v1 > v2 && v1 > v3) {                                      // Fast.h:275-285
    if (v1 >= S_(-1, 0) && v1 >= S_(-1, 1) && v1 >= S_(-1, 2) && v1 > S_(0, 2) &&
        v1 > S_(1, 2))
      return encode_fast(v1, x + 1, y);
    return 0;
  } else if (v2 > v3) {                                                 // Fast.h:287-296
    if (v2 >= S_(0, -1) && v2 >= S_(1, -1) && v2 > S_(2, -1) && v2 > S_(2, 0) &&
        v2 > S_(2, 1))
      return encode_fast(v2, x, y + 1);
    return 0;
  } else {                                                              // Fast.h:298-309
    if (v3 > S_(2, 0) && v3 > S_(2, 1) && v3 >= S_(0, 2) && v3 > S_(1, 2) && v3 > S_(2, 2))
      return encode_fast(v3, x + 1, y + 1);
    return 0;
  }
#undef S_
}

// Branch-free form of the same decision for one block whose 4x4 neighbourhood is already in
// registers: w0..w3 = the dwords S[y-1..y+2][x-1..x+2] (byte k = column x-1+k).  One LDS round trip
// instead of a chain of data-dependent reads; identical result to nms_block.
__device__ __forceinline__ uint32_t nms_block_regs(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, int x, int y) {
#define BYTE_(w, k) (((w) >> (8 * (k))) & 0xffu)
  const uint32_t v0 = BYTE_(w1, 1), v1 = BYTE_(w1, 2), v2 = BYTE_(w2, 1), v3 = BYTE_(w2, 2);
  const uint32_t a0 = BYTE_(w0, 0), a1 = BYTE_(w0, 1), a2 = BYTE_(w0, 2), a3 = BYTE_(w0, 3);
  const uint32_t b0 = BYTE_(w1, 0), b3 = BYTE_(w1, 3), c0 = BYTE_(w2, 0), c3 = BYTE_(w2, 3);
  const uint32_t d0 = BYTE_(w3, 0), d1 = BYTE_(w3, 1), d2 = BYTE_(w3, 2), d3 = BYTE_(w3, 3);
#undef BYTE_
  const bool s0 = (v0 > v1) & (v0 > v2) & (v0 > v3);                                   // Fast.h:264
  const bool s1 = !s0 & (v1 > v2) & (v1 > v3);                                         // Fast.h:275
  const bool s2 = !s0 & !s1 & (v2 > v3);                                               // Fast.h:287
  const bool ok0 = (v0 >= a0) & (v0 >= b0) & (v0 > c0) & (v0 >= a1) & (v0 >= a2);      // Fast.h:265-270
  const bool ok1 = (v1 >= a1) & (v1 >= a2) & (v1 >= a3) & (v1 > b3) & (v1 > c3);       // Fast.h:276-282
  const bool ok2 = (v2 >= b0) & (v2 >= c0) & (v2 > d0) & (v2 > d1) & (v2 > d2);        // Fast.h:288-293
  const bool ok3 = (v3 > d1) & (v3 > d2) & (v3 >= b3) & (v3 > c3) & (v3 > d3);         // Fast.h:299-305
  const bool any = (v0 | v1 | v2 | v3) != 0;                                           // Fast.h:237
  const uint32_t v = s0 ? v0 : (s1 ? v1 : (s2 ? v2 : v3));
  const bool ok = s0 ? ok0 : (s1 ? ok1 : (s2 ? ok2 : ok3));
  const int dx = (s0 | s2) ? 0 : 1, dy = (s0 | s1) ? 0 : 1;
  return (any & ok) ? encode_fast(v, x + dx, y + dy) : 0u;
}

// ---------------------------------------------------------------------------
// Orientation — reference Orb.h:310-387
// ---------------------------------------------------------------------------
// Half-width of the moment patch per |dy| (masks Orb.h:118-121, row macros
// Orb.h:163-178,208-220,238-250,271-286), packed 4 bits each, |dy| = 0..15.
__device__ __forceinline__ int patch_umax(int ady) {
  // {15,15,15,15,15,15,14,14,13,13,12,11,10,9,7,5}
  const uint64_t tab = 0x579ABCDDEEFFFFFFull;
  return (int)((tab >> (4 * ady)) & 0xf);
}

// The circle mask of the moment patch as dot-product weights, one row of the table per patch row r (dy = r - 15; row 31
// idle = all zero): dwords 0..7 hold 1 in every byte (dx = 4 k + b - 15) that belongs to the patch, dwords 8..15 hold
// |dx| there (Orb.h:118-126).  Built at compile time; the ORB lanes load their row (computing the 16 words per lane
// cost ~150 VALU per wave and workgroup: 8 % of k_gather_orb).
struct OrbMaskTab {
  uint32_t v[32][16];
};
static constexpr OrbMaskTab make_orb_mask_tab() {
  OrbMaskTab t{};
  constexpr int UMAX[16] = {15, 15, 15, 15, 15, 15, 14, 14, 13, 13, 12, 11, 10, 9, 7, 5};   // == patch_umax
  for (int r = 0; r < 31; r++) {
    const int dy = r - 15, u = UMAX[dy < 0 ? -dy : dy];
    for (int k = 0; k < 8; k++) {
      uint32_t m = 0, w = 0;
      for (int b = 0; b < 4; b++) {
        const int dx = 4 * k + b - 15, adx = dx < 0 ? -dx : dx;
        if (adx <= u) {
          m |= 1u << (8 * b);
          w |= (uint32_t)adx << (8 * b);
        }
      }
      t.v[r][k] = m;
      t.v[r][8 + k] = w;
    }
  }
  return t;
}

// NEON vrecpe.f32 (Orb.h:329): ARM ARM FPRecipEstimate, 8-bit, flush-to-zero.
__device__ __forceinline__ float vrecpe_f32(float f) {
  const uint32_t u = __float_as_uint(f), sign = u & 0x80000000u, e = (u >> 23) & 0xff,
                 m = u & 0x7fffffu;
  if (e == 0xff) return __uint_as_float(m ? 0x7fc00000u : sign);
  if (e == 0) return __uint_as_float(sign | 0x7f800000u);
  if (e >= 253) return __uint_as_float(sign);
  const uint32_t q2 = 2 * (256 + (m >> 15)) + 1;
  const uint32_t s = ((1u << 19) + q2) / (2 * q2);
  return __uint_as_float(sign | ((253 - e) << 23) | ((s - 256) << 15));
}

// The 8-bit estimate of FPRecipEstimate as a table: entry i (the top 8 mantissa bits) holds s - 256 with
// s = (2^19 + q) / (2 q), q = 2 (256 + i) + 1  — the integer division of vrecpe_f32, done at compile time.
struct VrecpeTab {
  uint8_t v[256];
};
static constexpr VrecpeTab make_vrecpe_tab() {
  VrecpeTab t{};
  for (int i = 0; i < 256; i++) {
    const uint32_t q2 = 2u * (256u + (uint32_t)i) + 1u;
    t.v[i] = (uint8_t)((((1u << 19) + q2) / (2u * q2)) - 256u);
  }
  return t;
}
// vrecpe_f32 with the table in LDS and the special cases as selects (no division, no branches)
template <class TAB>
__device__ __forceinline__ float vrecpe_f32_tab(float f, const TAB *tab) {
  const uint32_t u = __float_as_uint(f), sign = u & 0x80000000u, e = (u >> 23) & 0xff, m = u & 0x7fffffu;
  uint32_t r = sign | ((253u - e) << 23) | ((uint32_t)tab[m >> 15] << 15);
  r = e >= 253 ? sign : r;
  r = e == 0 ? (sign | 0x7f800000u) : r;
  r = e == 0xff ? (m ? 0x7fc00000u : sign) : r;
  return __uint_as_float(r);
}

// (m10, m01) -> angle bin 0..29.  Float ops are individually rounded (the
// library is built with -ffp-contract=off; ARMv7 NEON has no fused MAC here).
template <class RECIP>
__device__ __forceinline__ uint32_t angle_bin_with(int32_t x, int32_t y, RECIP recip) {
  const float xf = fabsf((float)x), yf = fabsf((float)y);      // Orb.h:318-322
  const float zmax = fmaxf(xf, yf), zmin = fminf(xf, yf);      // Orb.h:324-325
  const float z = __fmul_rn(zmin, recip(zmax));                // Orb.h:327-329
  const float c0 = (float)(256 * 14.999998);                   // Orb.h:336
  const float c1 = (float)(256 * 4.723436);                    // Orb.h:343
  const float c2 = (float)(256 * 1.266240);                    // Orb.h:344
  const float t1 = __fadd_rn(c1, __fmul_rn(c2, z));
  const float t3 = __fmul_rn(__fsub_rn(z, 1.0f), t1);
  const float af = __fmul_rn(z, __fsub_rn(c0, t3));            // Orb.h:345
  // vcvt.s32.f32 (Orb.h:348) truncates, saturates and maps NaN to 0 — exactly what v_cvt_i32_f32 does
  // (CDNA ISA: "out-of-range values saturate, NaN is converted to 0"; the (0,0) padding slots reach it as NaN)
  int32_t angle;
  asm("v_cvt_i32_f32 %0, %1" : "=v"(angle) : "v"(af));
  const uint32_t ax = x < 0 ? 0u - (uint32_t)x : (uint32_t)x;
  const uint32_t ay = y < 0 ? 0u - (uint32_t)y : (uint32_t)y;
  if (ax > ay) {                                               // Orb.h:355-364
    if ((x ^ y) < 0) angle = -angle;
    if (x < 0) angle += 256 * 60;
    else if (angle < 0) angle += 256 * 120;
  } else {                                                     // Orb.h:365-374
    if ((x ^ y) >= 0) angle = -angle;
    angle += (y >= 0) ? 256 * 30 : 256 * 90;
  }
  angle >>= 10;                                                // Orb.h:376
  if (!(0 <= angle && angle < 30)) angle = 0;                  // Orb.h:377-380
  return (uint32_t)angle;
}
// The same function with the integers doing what they can (Orb.h:318-380; results identical for every int32 pair):
//  * |float(x)| = float(|x|) (round-to-nearest-even is symmetric) and the int -> float conversion is monotone, so
//    zmax / zmin are the conversions of max / min of the ABSOLUTE INTEGERS — no float max / min (hipcc canonicalises
//    their operands with an extra instruction each), and |x| > |y| is the swap test anyway;
//  * vrecpe's argument is then a non-negative float with an exponent field of 0 (zero) or 127..158: of the ARM ARM
//    special cases only "zero -> +infinity" can occur;
//  * the octant logic as selects: negate iff [(x ^ y) < 0] == [|x| > |y|].
// `tab`: the 8-bit estimate table (VrecpeTab::v), in LDS or global memory.
template <class TAB>
__device__ __forceinline__ uint32_t angle_bin_fast(int32_t x, int32_t y, const TAB *tab) {
  const uint32_t ax = x < 0 ? 0u - (uint32_t)x : (uint32_t)x;
  const uint32_t ay = y < 0 ? 0u - (uint32_t)y : (uint32_t)y;
  const bool swap = ax > ay;
  const uint32_t imax = max(ax, ay), imin = min(ax, ay);
  const float zmax = (float)imax, zmin = (float)imin;          // v_cvt_f32_u32 (RNE), Orb.h:318-325
  const uint32_t u = __float_as_uint(zmax);
  uint32_t r = ((253u - (u >> 23)) << 23) | ((uint32_t)tab[(u >> 15) & 0xffu] << 15);   // FPRecipEstimate, Orb.h:329
  r = imax == 0 ? 0x7f800000u : r;
  const float z = __fmul_rn(zmin, __uint_as_float(r));         // Orb.h:327-329
  const float c0 = (float)(256 * 14.999998);                   // Orb.h:336
  const float c1 = (float)(256 * 4.723436);                    // Orb.h:343
  const float c2 = (float)(256 * 1.266240);                    // Orb.h:344
  const float t1 = __fadd_rn(c1, __fmul_rn(c2, z));
  const float t3 = __fmul_rn(__fsub_rn(z, 1.0f), t1);
  const float af = __fmul_rn(z, __fsub_rn(c0, t3));            // Orb.h:345
  int32_t angle;
  asm("v_cvt_i32_f32 %0, %1" : "=v"(angle) : "v"(af));        // vcvt.s32.f32 (Orb.h:348): truncates, saturates, NaN -> 0
  const bool opp = (x ^ y) < 0;
  angle = (opp == swap) ? -angle : angle;                      // Orb.h:355-374
  const int32_t add = swap ? (x < 0 ? 256 * 60 : (angle < 0 ? 256 * 120 : 0)) : (y >= 0 ? 256 * 30 : 256 * 90);
  angle = (angle + add) >> 10;                                 // Orb.h:376
  return (0 <= angle && angle < 30) ? (uint32_t)angle : 0u;    // Orb.h:377-380
}
__device__ __forceinline__ uint32_t angle_bin(int32_t x, int32_t y) {
  return angle_bin_with(x, y, [](float f) { return vrecpe_f32(f); });
}

// ---------------------------------------------------------------------------
// wave64 helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    uint32_t t = (uint32_t)__shfl_xor((int)v, o, 64);
    v = t > v ? t : v;
  }
  return v;
}
// number of set bits of a ballot below this lane
__device__ __forceinline__ int ballot_rank(uint64_t m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

}  // namespace pdev
