// pislam_fused_kernels.h — the measured path: fused detect + score + NMS per strip.
//
// One workgroup owns one horizontal strip (R rows x the full level width) of one
// level of one pyramid.  The image strip (+halo) is staged into LDS with 16-byte
// coalesced loads; everything up to the keypoint list happens out of LDS, so the
// image is read from HBM once and the reference's score map (`out`, Fast.h:54 /
// Fast.h:166) is never materialised in HBM:
//
//   phase A  cheap necessary test on every pixel (two adjacent compass points of
//            the Bresenham ring both darker / both brighter); survivors are
//            compacted per wave (ballot + mbcnt) into a wave-private LDS queue
//   phase B  whenever a queue holds >= 64 entries the wave pops 64 and runs the
//            full FAST-9 arc test on densely packed lanes (result-identical to
//            Fast.h:63-147); corners go to a second queue
//   phase C  same scheme for the 6x6 Harris score (Harris.h:80-248) -> LDS score tile
//   phase D  2x2-block NMS (Fast.h:228-312) on the LDS score tile; survivors are
//            written in block-raster order to the strip's slot of a staging buffer
//
// A second kernel (k_gather) turns per-strip counts into offsets (exclusive scan in
// strip order = reference order) and copies the staged keypoints to their final
// positions; k_orb then describes them.  Deterministic: no atomics-order dependence.
#pragma once
#include "pislam_dev.h"

namespace pf {

using namespace pdev;

constexpr int MAX_LEVELS = 16;
constexpr int WAVES = 4;               // 256 threads
constexpr int QCAP = 128;              // per-wave queue capacity (entries < 64 before a <=64 push)

struct FusedLevel {
  int w, h;          // level size
  int row0, col0;    // position in the stacked pyramid
  int R;             // strip height (even)
  int nstrips;       // strips in this level
  int strip0;        // index of the level's first strip within the pyramid
  int slot0;         // staging slot (in keypoints) of the level's first strip
  int nbx;           // 2x2 blocks per block-row
  int xend;          // one past the last classified column (Fast.h:61,149)
  int pitch;         // LDS tile pitch in bytes (multiple of 16)
};

struct FusedParams {
  int nlevels, strips_per_pyr, slots_per_pyr;
  int vstep, rows, border, thr;
  int32_t hthr;
  int batch;
  int dump_score;    // debug: also write the score tile to the HBM score map
  FusedLevel lv[MAX_LEVELS];
};

// candidate coordinates packed as x | (tile_row << 16)
__device__ __forceinline__ uint32_t pack_xy(int x, int r) { return (uint32_t)x | ((uint32_t)r << 16); }

// Necessary condition for a 9-arc: an arc of 9 ring positions contains two ADJACENT compass
// points (ring indices 1,5,9,13), so both must be dark (or both bright).
//   exists adjacent pair both > hi  <=>  min(max(p1,p9), max(p5,p13)) > hi
//   exists adjacent pair both < lo  <=>  max(min(p1,p9), min(p5,p13)) < lo
__device__ __forceinline__ bool fast_pretest(const uint8_t *c, int pitch, int thr) {
  const int v = c[0];
  const int p1 = c[-3 * pitch], p9 = c[3 * pitch], p5 = c[3], p13 = c[-3];
  const int mx = min(max(p1, p9), max(p5, p13));
  const int mn = max(min(p1, p9), min(p5, p13));
  return (mx > v + thr) | (mn < v - thr);
}

template <bool VEC16>
__global__ __launch_bounds__(256) void k_fused_strips(
    const FusedParams P, const uint8_t *__restrict__ pyramids, size_t pyr_stride,
    uint32_t *__restrict__ stage_kp, uint32_t *__restrict__ strip_count,
    uint8_t *__restrict__ score_dump, size_t score_stride) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  // XCD-aware mapping: workgroup b runs on XCD b%8; keep all strips of one pyramid on one XCD so
  // the halo rows shared by neighbouring strips are served by that XCD's L2.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int pyr = (slot / P.strips_per_pyr) * 8 + xcd;
  if (pyr >= P.batch) return;
  int s = slot % P.strips_per_pyr;
  int li = 0;
  while (li + 1 < P.nlevels && s >= P.lv[li + 1].strip0) li++;
  const FusedLevel L = P.lv[li];
  s -= L.strip0;
  const int B = P.border;
  const int ys = B + s * L.R;                       // first block-row y of the strip
  const int ye = min(ys + L.R, L.h - B);            // one past the last row owned
  const int pitch = L.pitch;
  const int trows = L.R + 10;                       // image tile rows  [ys-4, ys+R+6)
  uint8_t *tile = smem;
  uint8_t *sc = smem + trows * pitch;               // score tile rows  [ys-1, ys+R+2)
  uint32_t *queues = (uint32_t *)(sc + (L.R + 3) * pitch);

  const uint8_t *im = pyramids + (size_t)pyr * pyr_stride + (size_t)L.row0 * P.vstep + L.col0;
  const int tid = threadIdx.x;

  // ---- stage the image rows [ys-4, min(ye+6, h)) and clear the score tile ---------------
  {
    const int y_lo = ys - 4;
    const int nrows = min(ye + 6, L.h) - y_lo;
    if (VEC16) {
      const int vpr = pitch >> 4;                   // 16-byte vectors per row
      // bytes of this pyramid's buffer that may be read (the tile can overhang the last image row
      // when col0 + pitch > vstep: flat addressing like the reference, but never past the buffer)
      const ptrdiff_t lim = (ptrdiff_t)P.rows * P.vstep - ((ptrdiff_t)L.row0 * P.vstep + L.col0);
      for (int i = tid; i < nrows * vpr; i += 256) {
        const int r = i / vpr, v = i - r * vpr;
        const ptrdiff_t off = (ptrdiff_t)(y_lo + r) * P.vstep + 16 * v;
        uint4 d;
        if (off + 16 <= lim) {
          d = *(const uint4 *)(im + off);
        } else {
          uint8_t t[16];
          for (int k = 0; k < 16; k++) t[k] = off + k < lim ? im[off + k] : (uint8_t)0;
          d = *(const uint4 *)t;
        }
        *(uint4 *)(tile + r * pitch + 16 * v) = d;
      }
    } else {
      const ptrdiff_t lim = (ptrdiff_t)P.rows * P.vstep - ((ptrdiff_t)L.row0 * P.vstep + L.col0);
      for (int i = tid; i < nrows * pitch; i += 256) {
        const int r = i / pitch, cx = i - r * pitch;
        const ptrdiff_t off = (ptrdiff_t)(y_lo + r) * P.vstep + cx;
        tile[r * pitch + cx] = off < lim ? im[off] : (uint8_t)0;
      }
    }
    const int nz = ((L.R + 3) * pitch) >> 4;
    for (int i = tid; i < nz; i += 256) ((uint4 *)sc)[i] = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();

  const int lane = lane_id();
  const int wave = tid >> 6;
  uint32_t *qf = queues + wave * (2 * QCAP);        // FAST candidates
  uint32_t *qh = qf + QCAP;                         // corners awaiting their Harris score
  int nf = 0, nh = 0;                               // wave-uniform queue fills
  const int thr = P.thr;
  const bool wmod = (L.w & 15) != 0;

  // score rows r = 0 .. R+2  <->  level rows ys-1+r ; image tile row of level row y is y-(ys-4)
  auto harris_batch = [&](bool valid, uint32_t e) {
    if (valid) {
      const int x = e & 0xffff, r = e >> 16;
      sc[r * pitch + x] = harris_score(tile + (r + 3) * pitch + x, pitch, P.hthr);
    }
  };
  auto fast_batch = [&](bool valid, uint32_t e) {
    bool corner = false;
    const int x = e & 0xffff, r = e >> 16;
    if (valid) corner = fast9(tile + (r + 3) * pitch + x, pitch, thr);
    // Fast.h:172: only x < w-B is scored; over-classified columns keep 0xff
    const bool toh = corner && x < L.w - B;
    if (corner && !toh) sc[r * pitch + x] = 0xff;
    const uint64_t m = __ballot(toh);
    if (m) {
      if (toh) qh[nh + ballot_rank(m)] = e;
      nh += __popcll(m);
      if (nh >= 64) {
        nh -= 64;
        harris_batch(true, qh[nh + lane]);
      }
    }
  };

  const int r_lo = (ys - 1 < B) ? 1 : 0;                          // rows above B are never classified
  const int r_hi = min(ye + 2, L.h - B) - (ys - 1);               // exclusive
  for (int r = r_lo + wave; r < r_hi; r += WAVES) {
    const uint8_t *trow = tile + (r + 3) * pitch;
    for (int cx = B; cx < L.xend; cx += 64) {
      const int x = cx + lane;
      bool cand = false;
      if (x < L.xend && !(wmod && (x == L.w || x == L.w + 1))) cand = fast_pretest(trow + x, pitch, thr);
      const uint64_t m = __ballot(cand);
      if (m) {
        if (cand) qf[nf + ballot_rank(m)] = pack_xy(x, r);
        nf += __popcll(m);
        if (nf >= 64) {
          nf -= 64;
          fast_batch(true, qf[nf + lane]);
        }
      }
    }
  }
  if (nf > 0) fast_batch(lane < nf, lane < nf ? qf[lane] : 0u);
  if (nh > 0) harris_batch(lane < nh, lane < nh ? qh[lane] : 0u);
  __syncthreads();

  if (P.dump_score) {   // debug / parity hook: rows this strip owns, [ys, ye)
    uint8_t *dst = score_dump + (size_t)pyr * score_stride + (size_t)L.row0 * P.vstep + L.col0;
    for (int i = tid; i < (ye - ys) * pitch; i += 256) {
      const int r = i / pitch, x = i - r * pitch;
      if (x >= B && x < max(L.xend, wmod ? L.w + 2 : 0)) dst[(ptrdiff_t)(ys + r) * P.vstep + x] = sc[(r + 1) * pitch + x];
    }
  }

  // ---- phase D: NMS, block-raster order.  Pass 0 counts per block-row, pass 1 scatters. ----
  __shared__ uint32_t rowcnt[64];
  const int nbr = (ye - ys + 1) >> 1;               // block rows in this strip
  for (int br = wave; br < nbr; br += WAVES) {
    const uint8_t *srow = sc + (2 * br + 1) * pitch;
    uint32_t cnt = 0;
    for (int bx0 = 0; bx0 < L.nbx; bx0 += 64) {
      const int bx = bx0 + lane;
      uint32_t res = 0;
      if (bx < L.nbx) res = nms_block(srow + B + 2 * bx, pitch, B + 2 * bx, ys + 2 * br);
      cnt += __popcll(__ballot(res != 0));
    }
    if (lane == 0) rowcnt[br] = cnt;
  }
  __syncthreads();
  const size_t strip_slot = (size_t)pyr * P.slots_per_pyr + L.slot0 + (size_t)s * (L.R >> 1) * L.nbx;
  const uint32_t add_xy = ((uint32_t)L.col0 << 12) | (uint32_t)L.row0;      // README.md:78
  for (int br = wave; br < nbr; br += WAVES) {
    uint32_t off = 0;
    for (int k = 0; k < br; k++) off += rowcnt[k];
    if (rowcnt[br] == 0) continue;
    const uint8_t *srow = sc + (2 * br + 1) * pitch;
    for (int bx0 = 0; bx0 < L.nbx; bx0 += 64) {
      const int bx = bx0 + lane;
      uint32_t res = 0;
      if (bx < L.nbx) res = nms_block(srow + B + 2 * bx, pitch, B + 2 * bx, ys + 2 * br);
      const uint64_t m = __ballot(res != 0);
      if (res) stage_kp[strip_slot + off + ballot_rank(m)] = res + add_xy;
      off += __popcll(m);
    }
  }
  if (tid == 0) {
    uint32_t tot = 0;
    for (int k = 0; k < nbr; k++) tot += rowcnt[k];
    strip_count[(size_t)pyr * P.strips_per_pyr + L.strip0 + s] = tot;
  }
}

// One workgroup per pyramid: exclusive scan of the strip counts in strip order (= level order,
// then top-to-bottom = the reference's push_back order), copy staged keypoints to their final
// positions, publish the total.
__global__ __launch_bounds__(256) void k_gather(const FusedParams P,
                                                const uint32_t *__restrict__ stage_kp,
                                                const uint32_t *__restrict__ strip_count,
                                                uint32_t *__restrict__ kp, size_t kp_stride,
                                                uint32_t cap, uint32_t *__restrict__ counts) {
  extern __shared__ uint32_t soff[];                // strips_per_pyr + 1
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t carry;
  const int pyr = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int S = P.strips_per_pyr;
  const uint32_t *cnt = strip_count + (size_t)pyr * S;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < S; base += 256) {
    const int i = base + tid;
    const uint32_t v = i < S ? cnt[i] : 0;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t t = (uint32_t)__shfl_up((int)incl, d, 64);
      if (lane >= d) incl += t;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t pre = carry;
    for (int w = 0; w < wv; w++) pre += wsum[w];
    if (i < S) soff[i] = pre + incl - v;
    __syncthreads();
    if (tid == 255) carry = pre + incl;
    __syncthreads();
  }
  if (tid == 0) {
    soff[S] = carry;
    counts[pyr] = carry;
  }
  __syncthreads();
  // one wave per strip
  for (int st = wv; st < S; st += 4) {
    const uint32_t n = soff[st + 1] - soff[st];
    if (n == 0) continue;
    int li = 0;
    while (li + 1 < P.nlevels && st >= P.lv[li + 1].strip0) li++;
    const FusedLevel &L = P.lv[li];
    const size_t slot = (size_t)pyr * P.slots_per_pyr + L.slot0 + (size_t)(st - L.strip0) * (L.R >> 1) * L.nbx;
    for (uint32_t k = lane; k < n; k += 64) {
      const uint32_t pos = soff[st] + k;
      if (pos < cap) kp[(size_t)pyr * kp_stride + pos] = stage_kp[slot + k];
    }
  }
}

}  // namespace pf
