// pislam_stage_kernels.h — one HIP kernel per reference entry point.
//
// These back the drop-in 4-call API (pislam_fast_detect / _score_harris /
// _extract / _orb_compute) and, with blockIdx.z = pyramid index, the first
// (staged) batch pipeline.  All of them are deterministic: ordered outputs come
// from count -> exclusive scan -> scatter, never from atomics-order.
#pragma once
#include "pislam_dev.h"

namespace pk {

using namespace pdev;

// ---------------------------------------------------------------------------
// K1 fast_detect — reference Fast.h:54-158
// grid (tiles_x, tiles_y, batch), 256 threads; tile 64 x 16 px, image tile with
// a 3 px halo staged in LDS.
// ---------------------------------------------------------------------------
constexpr int DT_W = 64, DT_H = 16, DT_PITCH = 72;   // 64 + 6 halo, padded to 72

__global__ __launch_bounds__(256) void k_fast_detect(
    const uint8_t *__restrict__ img, uint8_t *__restrict__ out, int vstep, size_t pyr_stride,
    int border, int width, int height, int thr, int xend) {
  __shared__ uint8_t tile[(DT_H + 6) * DT_PITCH];
  const uint8_t *im = img + (size_t)blockIdx.z * pyr_stride;
  uint8_t *o = out + (size_t)blockIdx.z * pyr_stride;
  const int x0 = border + blockIdx.x * DT_W, y0 = border + blockIdx.y * DT_H;
  const int yend = height - border;
  // stage rows y0-3 .. y0+DT_H+2, cols x0-3 .. x0+DT_W+2 (only what a classified
  // pixel can touch: cols < xend+3, rows < yend+3)
  for (int i = threadIdx.x; i < (DT_H + 6) * (DT_W + 6); i += 256) {
    const int r = i / (DT_W + 6), c = i - r * (DT_W + 6);
    const int gy = y0 - 3 + r, gx = x0 - 3 + c;
    uint8_t v = 0;
    if (gy < yend + 3 && gx < xend + 3) v = im[(ptrdiff_t)gy * vstep + gx];
    tile[r * DT_PITCH + c] = v;
  }
  __syncthreads();
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const bool wmod = (width % 16) != 0;
#pragma unroll
  for (int j = 0; j < DT_H / 4; j++) {
    const int r = ty + 4 * j;
    const int x = x0 + tx, y = y0 + r;
    if (x < xend && y < yend) {
      // Fast.h:153-156: out[y][width], out[y][width+1] end up 0 when width%16 != 0
      if (!(wmod && (x == width || x == width + 1))) {
        const bool f = fast9(&tile[(r + 3) * DT_PITCH + tx + 3], DT_PITCH, thr);
        o[(ptrdiff_t)y * vstep + x] = f ? 0xff : 0x00;
      }
    }
    if (wmod && blockIdx.x == 0 && tx < 2 && y < yend) o[(ptrdiff_t)y * vstep + width + tx] = 0;
  }
}

// ---------------------------------------------------------------------------
// K2 fast_score_harris — reference Fast.h:166-180 + Harris.h:80-248
// Same tiling; non-zero pixels of the tile are compacted into an LDS list so
// that the (expensive, rare) Harris evaluation runs on densely packed lanes.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_harris_score(
    const uint8_t *__restrict__ img, uint8_t *__restrict__ out, int vstep, size_t pyr_stride,
    int border, int width, int height, int32_t threshold) {
  __shared__ uint32_t list[DT_W * DT_H];
  __shared__ int count;
  const uint8_t *im = img + (size_t)blockIdx.z * pyr_stride;
  uint8_t *o = out + (size_t)blockIdx.z * pyr_stride;
  if (threadIdx.x == 0) count = 0;
  __syncthreads();
  const int x0 = border + blockIdx.x * DT_W, y0 = border + blockIdx.y * DT_H;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < DT_H / 4; j++) {
    const int x = x0 + tx, y = y0 + ty + 4 * j;
    const bool nz = x < width - border && y < height - border && o[(ptrdiff_t)y * vstep + x] != 0;
    const uint64_t m = __ballot(nz);
    int base = 0;
    if (m) {
      if (lane_id() == 0) base = atomicAdd(&count, __popcll(m));
      base = __shfl(base, 0, 64);
      if (nz) list[base + ballot_rank(m)] = (uint32_t)x | ((uint32_t)y << 16);
    }
  }
  __syncthreads();
  const int n = count;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int x = list[i] & 0xffff, y = list[i] >> 16;
    o[(ptrdiff_t)y * vstep + x] = harris_score(im + (ptrdiff_t)y * vstep + x, vstep, threshold);
  }
}

// harrisScoreSobel on an explicit point list (Harris.h:80)
__global__ void k_harris_points(const uint8_t *__restrict__ img, int vstep,
                                const uint32_t *__restrict__ pts, int n, int32_t threshold,
                                uint8_t *__restrict__ scores) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = decode_x(pts[i]), y = decode_y(pts[i]);
  scores[i] = harris_score(img + (ptrdiff_t)y * vstep + x, vstep, threshold);
}

// ---------------------------------------------------------------------------
// K3 fast_extract without buckets — reference Fast.h:196-355, logBucketSize=0.
// One wave per 2-row block-row.  Pass 0 counts survivors per block-row, a scan
// turns counts into offsets, pass 1 recomputes and scatters in raster order
// (= the reference's push_back order).
// ---------------------------------------------------------------------------
template <bool EMIT>
__global__ __launch_bounds__(256) void k_nms_rows(
    const uint8_t *__restrict__ score, int vstep, size_t pyr_stride, int border, int width,
    int height, int nrows, uint32_t *__restrict__ rowcount, size_t rc_stride,
    const uint32_t *__restrict__ rowoff, uint32_t *__restrict__ kp, size_t kp_stride, uint32_t cap,
    uint32_t add_xy) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const int b = blockIdx.z;
  const uint8_t *s = score + (size_t)b * pyr_stride;
  const int y = border + 2 * row;
  const int lane = lane_id();
  const int nbx = (width - 2 * border + 1) / 2;   // blocks x = border, border+2, ... < width-border
  uint32_t running = EMIT ? rowoff[(size_t)b * rc_stride + row] : 0;
  for (int bx = lane; bx - lane < nbx; bx += 64) {
    uint32_t r = 0;
    if (bx < nbx) {
      const int x = border + 2 * bx;
      r = nms_block(s + (ptrdiff_t)y * vstep + x, vstep, x, y);
    }
    const uint64_t m = __ballot(r != 0);
    if (EMIT) {
      const uint32_t pos = running + ballot_rank(m);
      if (r != 0 && pos < cap) kp[(size_t)b * kp_stride + pos] = r + add_xy;
    }
    running += __popcll(m);
  }
  if (!EMIT && lane == 0) rowcount[(size_t)b * rc_stride + row] = running;
}

// Exclusive scan of n counts (one workgroup per pyramid), offset by and
// accumulated into total[b]:  off[i] = total[b] + sum(cnt[0..i)),  total[b] += sum.
__global__ __launch_bounds__(256) void k_scan_counts(const uint32_t *__restrict__ cnt,
                                                     uint32_t *__restrict__ off, int n,
                                                     size_t stride, uint32_t *__restrict__ total) {
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t carry;
  const int b = blockIdx.x;
  const uint32_t *c = cnt + (size_t)b * stride;
  uint32_t *o = off + (size_t)b * stride;
  if (threadIdx.x == 0) carry = total[b];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int base = 0; base < n; base += 256) {
    const int i = base + threadIdx.x;
    const uint32_t v = i < n ? c[i] : 0;
    uint32_t incl = v;   // inclusive scan inside the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t t = (uint32_t)__shfl_up((int)incl, d, 64);
      if (lane >= d) incl += t;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t pre = carry;
    for (int w = 0; w < wv; w++) pre += wsum[w];
    if (i < n) o[i] = pre + incl - v;
    __syncthreads();
    if (threadIdx.x == 255) carry = pre + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) total[b] = carry;
}

// ---------------------------------------------------------------------------
// K3 fast_extract with buckets — reference Fast.h:196-355, logBucketSize>0.
// A "cell" is one bucket (2^lbs columns of block origins) x one flush interval
// (2^lbs rows).  One wave per cell: survivors of its 2x2 blocks go to an LDS
// array, the bucketLimit largest packed words are selected by repeated
// wave-max and stored ASCENDING (the order the reference's insertion sort
// leaves them in, Fast.h:321-341); cells are emitted in (cell-row, bucket)
// order (Fast.h:211-226,345-352).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_nms_cells(
    const uint8_t *__restrict__ score, int vstep, size_t pyr_stride, int border, int width,
    int height, int lbs, int limit, int ncx, uint32_t *__restrict__ cellcount,
    uint32_t *__restrict__ cellkp, size_t cc_stride) {
  extern __shared__ uint32_t cand[];
  const int cx = blockIdx.x, cy = blockIdx.y, b = blockIdx.z;
  const uint8_t *s = score + (size_t)b * pyr_stride;
  const int bs = 1 << lbs, hb = bs >> 1;       // blocks per cell side
  const int lane = threadIdx.x;
  const int xc0 = border + cx * bs, yc0 = border + cy * bs;
  const int nblk = hb * hb;
  for (int i = lane; i < nblk; i += 64) {
    const int by = i / hb, bx = i - by * hb;
    const int x = xc0 + 2 * bx, y = yc0 + 2 * by;
    uint32_t r = 0;
    if (x < width - border && y < height - border)
      r = nms_block(s + (ptrdiff_t)y * vstep + x, vstep, x, y);
    cand[i] = r;
  }
  __syncthreads();
  const size_t cell = (size_t)b * cc_stride + (size_t)cy * ncx + cx;
  int nf = 0;
  for (int k = 0; k < limit; k++) {
    uint32_t best = 0;
    for (int i = lane; i < nblk; i += 64) best = cand[i] > best ? cand[i] : best;
    best = wave_max_u32(best);
    if (best == 0) break;
    for (int i = lane; i < nblk; i += 64)
      if (cand[i] == best) cand[i] = 0;
    __syncthreads();
    if (lane == 0) cellkp[cell * limit + k] = best;   // descending; reversed on emit
    nf++;
  }
  if (lane == 0) cellcount[cell] = nf;
}

__global__ __launch_bounds__(256) void k_emit_cells(
    const uint32_t *__restrict__ cellcount, const uint32_t *__restrict__ celloff,
    const uint32_t *__restrict__ cellkp, int ncells, int limit, size_t cc_stride,
    uint32_t *__restrict__ kp, size_t kp_stride, uint32_t cap, uint32_t add_xy) {
  const int b = blockIdx.z;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= ncells * limit) return;
  const int cell = i / limit, k = i - cell * limit;
  const size_t c = (size_t)b * cc_stride + cell;
  const uint32_t n = cellcount[c];
  if ((uint32_t)k >= n) return;
  const uint32_t pos = celloff[c] + k;                       // ascending position k
  if (pos < cap) kp[(size_t)b * kp_stride + pos] = cellkp[c * limit + (n - 1 - k)] + add_xy;
}

// ---------------------------------------------------------------------------
// K4 ORB — reference Orb.h:80-308 (moments), 310-387 (angle), Brief.h (bits),
// Orb.h:396-441 (driver).  One wave per keypoint.
//   lane l: column dx = (l & 31) - 15, rows dy = -15 + (l >> 5) + 2j.
// MODE 0: full orbCompute (descriptors)          MODE 1: centroids only
// MODE 2: briefDescribe with given rotations
// ---------------------------------------------------------------------------
// g_brief_tab (uint32[30*256], packed dx0|dy0<<8|dx1<<16|dy1<<24) is defined by the including TU.

template <int MODE>
__global__ __launch_bounds__(256) void k_orb(
    const uint8_t *__restrict__ img, int vstep, size_t pyr_stride,
    const uint32_t *__restrict__ pts, size_t pts_stride, const uint32_t *__restrict__ counts,
    uint32_t n_fixed, uint32_t cap, int words, uint32_t *__restrict__ desc, size_t desc_stride,
    int32_t *__restrict__ cen, const uint8_t *__restrict__ rots) {
  const int b = blockIdx.z;
  const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6);
  uint32_t n = counts ? counts[b] : n_fixed;
  if (n > cap) n = cap;
  if (i >= n) return;
  const uint8_t *im = img + (size_t)b * pyr_stride;
  const uint32_t p = pts[(size_t)b * pts_stride + i];
  const int x = decode_x(p), y = decode_y(p);
  const uint8_t *c = im + (ptrdiff_t)y * vstep + x;
  const int lane = lane_id();
  uint32_t rot;
  if (MODE != 2) {
    const int dx = (lane & 31) - 15;
    int m10 = 0, m01 = 0;
#pragma unroll 4
    for (int j = 0; j < 16; j++) {
      const int dy = -15 + (lane >> 5) + 2 * j;
      if (dy <= 15) {
        const int ady = dy < 0 ? -dy : dy, adx = dx < 0 ? -dx : dx;
        if (adx <= patch_umax(ady)) {
          const int v = c[dy * vstep + dx];
          m10 += dx * v;
          m01 += dy * v;
        }
      }
    }
    m10 = wave_sum(m10);
    m01 = wave_sum(m01);
    if (MODE == 1) {
      if (lane == 0) {
        const size_t g = (size_t)(i >> 2) * 8 + (i & 3);   // Orb.h:298-306 grouped layout
        cen[g] = m10;
        cen[g + 4] = m01;
      }
      return;
    }
    rot = angle_bin_fast(m10, m01, ::g_vrecpe_tab.v);
  } else {
    rot = rots[i];
    if (rot >= 30) return;   // Brief.h:641-732: switch without default writes nothing
  }
  uint32_t *d = desc + (size_t)b * desc_stride + (size_t)i * words;
  const uint32_t *tab = g_brief_tab + rot * 256;
  for (int r = 0; r * 2 < words; r++) {
    const uint32_t e = tab[r * 64 + lane];
    const int dx0 = (int8_t)(e & 0xff), dy0 = (int8_t)((e >> 8) & 0xff);
    const int dx1 = (int8_t)((e >> 16) & 0xff), dy1 = (int8_t)(e >> 24);
    const uint32_t a = c[dy0 * vstep + dx0], bb = c[dy1 * vstep + dx1];
    const uint64_t m = __ballot(a < bb);                    // Brief.h:52
    if (lane == 0) {
      d[2 * r] = (uint32_t)m;
      if (2 * r + 1 < words) d[2 * r + 1] = (uint32_t)(m >> 32);
    }
  }
}

// Orb.h:310-387 on the grouped centroid layout: slot s of group g reads
// xys[8g+s], xys[8g+4+s] and writes angles[4g+s].
__global__ void k_angles(const int32_t *__restrict__ xys, int nslots, uint8_t *__restrict__ ang) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nslots) return;
  const int g = i >> 2, s = i & 3;
  ang[i] = (uint8_t)angle_bin_fast(xys[8 * g + s], xys[8 * g + 4 + s], ::g_vrecpe_tab.v);
}

}  // namespace pk
