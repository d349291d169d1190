// band_probe.hip — exploration for DESIGN.md section 8 ("waves that roll down a column band"): the classification
// phases of the ORB front-end (SAD prefilter -> exact compass pretest -> FAST-9 -> Harris score) with ONE WAVE per work item, a
// rolling window of image rows in the wave's private LDS and FIFO candidate queues that stay alive over the whole
// segment — no workgroup barriers, partly filled batches only when a queue entry is about to lose its image rows
// and at the end of a segment.  Counts the FAST corners per item and checks pyramid 0 against a textbook FAST-9 on
// the CPU.  NOT part of the product.
//
//   hipcc --offload-arch=gfx950 -O3 -I ../../pislam_amd/csrc band_probe.hip -o band_probe
//   band_probe pyramids.raw [batch] [segment_rows]      (pyramids.raw: uint8 [batch][2210][640], VGA level table)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "pislam_dev.h"

typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4 lds_u4;
typedef unsigned short us2 __attribute__((ext_vector_type(2)));

constexpr int P = 160;                 // LDS row pitch: 16 halo + 128 owned + 16 halo columns
constexpr int NR = 32, NDUP = 7, NSLOT = NR + NDUP;   // ring of 32 image rows; rows with slot < 7 are duplicated behind
                                                     // the ring so that rows c-3 .. c+4 are contiguous for every centre row
constexpr int QG = 128, QF = 256, QC = 128;   // FIFO capacities (dwords): groups, FAST candidates, corners
constexpr int LDS_BYTES = NSLOT * P + 4 * (QG + QF + QC);
constexpr int HTHR = 1 << 15;
constexpr int VSTEP = 640, ROWS = 2210, THR = 20, B = 16;

struct Item {
  int row0, h;       // level position / height
  int cx0, cx1;      // owned classified columns (level-relative), cx1 - cx0 <= 128, multiple of 4
  int y0, y1;        // classified rows (level-relative)
  int xscore;        // w - B: corners at / beyond it keep 0xff (Fast.h:172)
};

__device__ __forceinline__ us2 as_us2(uint32_t v) { return __builtin_bit_cast(us2, v); }
__device__ __forceinline__ uint32_t as_u32(us2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ void pretest_pk(uint32_t c, uint32_t u, uint32_t d, uint32_t l, uint32_t r, uint32_t t2,
                                           uint32_t &bright, uint32_t &dark) {
  const us2 C = as_us2(c), U = as_us2(u), D = as_us2(d), Lf = as_us2(l), Rt = as_us2(r), T = as_us2(t2);
  const us2 a = __builtin_elementwise_min(__builtin_elementwise_max(U, D), __builtin_elementwise_max(Lf, Rt));
  const us2 b = __builtin_elementwise_max(__builtin_elementwise_min(U, D), __builtin_elementwise_min(Lf, Rt));
  const us2 hi = C + T, lo = C - T;
  bright = as_u32(hi - a);
  dark = as_u32(b - lo);
}
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__global__ __launch_bounds__(64) void k_bands(const Item *__restrict__ items, const uint8_t *__restrict__ pyramids,
                                              uint32_t *__restrict__ out_count, uint32_t *__restrict__ out_sum, int nitems,
                                              uint8_t *__restrict__ score_map) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  lds_u8 *ring = (lds_u8 *)smem;
  lds_u32 *qg = (lds_u32 *)(smem + NSLOT * P);
  lds_u32 *qf = qg + QG;
  lds_u32 *qc = qf + QF;
  const Item it = items[blockIdx.x];
  const int lane = threadIdx.x;
  const uint8_t *im = pyramids + (size_t)blockIdx.y * ROWS * VSTEP + (size_t)it.row0 * VSTEP;
  const int xs0 = it.cx0 - 16;                       // first staged column
  // staging: 8 rows x 10 vectors per chunk, vector v = lane + 64 k -> (row v / 10, column v % 10)
  int srow[2], scol[2];
  bool son[2];
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int v = lane + 64 * k;
    srow[k] = v / 10;
    scol[k] = v - 10 * srow[k];
    son[k] = v < 80;
  }
  auto gaddr = [&](int yy, int col) { return im + (size_t)yy * VSTEP + min(xs0 + 16 * col, VSTEP - 16); };
  auto park = [&](int yy, int col, u32x4 d) {        // image row yy -> its ring slot (and the duplicate)
    const int s = yy & (NR - 1);
    *(lds_u4 *)(ring + s * P + 16 * col) = d;
    if (s < NDUP) *(lds_u4 *)(ring + (s + NR) * P + 16 * col) = d;
  };
  auto rowptr = [&](int y) -> const lds_u8 * {       // row y with rows y-3 .. y+3 contiguous around it
    int c = y & (NR - 1);
    c = c < 3 ? c + NR : c;
    return ring + c * P - xs0;
  };
  uint32_t ncorner = 0, csum = 0;
  uint32_t head_g = 0, tail_g = 0, head_f = 0, tail_f = 0, head_c = 0, tail_c = 0;   // wave-uniform FIFO cursors (free running)
  uint32_t nscored = 0, nbatch_h = 0, nbatch_f = 0, nbatch_p = 0;
  const uint32_t t2 = (uint32_t)THR * 0x00010001u;

  // Harris for 64 corners (pdev::harris_score_mm on the 8 x 8 window rows y-3 .. y+4); Fast.h:172: only x < w - B
  auto harris_batch = [&](bool valid, uint32_t e) {
    const int x = e & 0xfff, y = (e >> 12) & 0xfff;
    uint8_t sc = 0;
    if (valid) sc = x < it.xscore ? pdev::harris_score_mm((const pdev::lds_byte *)(rowptr(y) - 3 * P + x - 3), P, HTHR) : (uint8_t)0xff;
    nscored += (uint32_t)__popcll(__ballot(sc != 0));
    if (sc != 0) csum += (e + sc) * 2654435761u;
    if (score_map && blockIdx.y == 0 && valid) score_map[(size_t)(it.row0 + y) * VSTEP + x] = sc ? sc : (uint8_t)1;   // 1 = corner below the threshold
  };
  auto pop_corners = [&](bool all) {
    while (tail_c - head_c >= 64u || (all && tail_c != head_c)) {
      const uint32_t n = min(64u, tail_c - head_c);
      lds_wait();
      nbatch_h++;
      harris_batch((uint32_t)lane < n, qc[(head_c + min((uint32_t)lane, n - 1)) & (QC - 1)]);
      head_c += n;
    }
  };
  auto fast_batch = [&](bool valid, uint32_t e) {
    bool corner = false;
    const int x = e & 0xfff, y = (e >> 12) & 0xfff;
    if (valid) corner = pdev::fast9_mm(rowptr(y) + x, P, THR);
    const uint64_t m = __ballot(corner);
    if (m == 0) return;
    ncorner += (uint32_t)__popcll(m);
    if (corner) qc[(tail_c + pdev::ballot_rank(m)) & (QC - 1)] = e;
    tail_c += (uint32_t)__popcll(m);
    pop_corners(false);
  };
  auto pop_fast = [&](bool all) {
    while (tail_f - head_f >= 64u || (all && tail_f != head_f)) {
      const uint32_t n = min(64u, tail_f - head_f);
      lds_wait();
      nbatch_f++;
      fast_batch((uint32_t)lane < n, qf[(head_f + min((uint32_t)lane, n - 1)) & (QF - 1)]);
      head_f += n;
    }
  };
  auto pretest_batch = [&](bool valid, uint32_t key) {   // key: x | row << 12 of a 4-pixel group
    const int x0 = key & 0xfff, y = (key >> 12) & 0xfff;
    // candidates carry (their row - the first row of this batch) in bits 24..31: the FIFOs downstream are ordered by
    // pretest batch, not by row, and "row - delta" of a FIFO's head entry is a lower bound of all its pending rows
    key |= (uint32_t)(y - __builtin_amdgcn_readfirstlane(y)) << 24;
    const lds_u8 *trow = rowptr(y);
    const uint32_t wc = *(const lds_u32 *)(trow + x0);
    const uint32_t wl = *(const lds_u32 *)(trow + x0 - 4);
    const uint32_t wr = *(const lds_u32 *)(trow + x0 + 4);
    const uint32_t wu = *(const lds_u32 *)(trow + x0 - 3 * P);
    const uint32_t wd = *(const lds_u32 *)(trow + x0 + 3 * P);
    uint32_t be, de, bo, dd;
    pretest_pk(__builtin_amdgcn_perm(0, wc, 0x0c020c00u), __builtin_amdgcn_perm(0, wu, 0x0c020c00u),
               __builtin_amdgcn_perm(0, wd, 0x0c020c00u), __builtin_amdgcn_perm(wc, wl, 0x0c030c01u),
               __builtin_amdgcn_perm(wr, wc, 0x0c050c03u), t2, be, de);
    pretest_pk(__builtin_amdgcn_perm(0, wc, 0x0c030c01u), __builtin_amdgcn_perm(0, wu, 0x0c030c01u),
               __builtin_amdgcn_perm(0, wd, 0x0c030c01u), __builtin_amdgcn_perm(wc, wl, 0x0c040c02u),
               __builtin_amdgcn_perm(wr, wc, 0x0c060c04u), t2, bo, dd);
    const uint32_t re = be | de, ro = bo | dd;
    const uint32_t fe = valid ? re & 0x80008000u : 0u, fo = valid ? ro & 0x80008000u : 0u;
    if (__ballot((fe | fo) != 0) == 0) return;
    const uint64_t m0 = __ballot((fe & 0x8000u) != 0), m1 = __ballot((fo & 0x8000u) != 0);
    const uint64_t m2 = __ballot((int32_t)fe < 0), m3 = __ballot((int32_t)fo < 0);
    {
      uint32_t at = tail_f;
      if (fe & 0x8000u) qf[(at + pdev::ballot_rank(m0)) & (QF - 1)] = key;
      at += __popcll(m0);
      if (fo & 0x8000u) qf[(at + pdev::ballot_rank(m1)) & (QF - 1)] = key + 1;
      tail_f = at + __popcll(m1);
    }
    pop_fast(false);
    {
      uint32_t at = tail_f;
      if ((int32_t)fe < 0) qf[(at + pdev::ballot_rank(m2)) & (QF - 1)] = key + 2;
      at += __popcll(m2);
      if ((int32_t)fo < 0) qf[(at + pdev::ballot_rank(m3)) & (QF - 1)] = key + 3;
      tail_f = at + __popcll(m3);
    }
    pop_fast(false);
  };
  auto pop_groups = [&](bool all) {
    while (tail_g - head_g >= 64u || (all && tail_g != head_g)) {
      const uint32_t n = min(64u, tail_g - head_g);
      lds_wait();
      nbatch_p++;
      pretest_batch((uint32_t)lane < n, qg[(head_g + min((uint32_t)lane, n - 1)) & (QG - 1)]);
      head_g += n;
    }
  };
  auto oldest_row = [&](lds_u32 *q, uint32_t head, int mask) -> int {   // lower bound of the pending rows of a FIFO
    lds_wait();
    const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)q[head & mask]);
    return (int)((e >> 12) & 0xfff) - (int)(e >> 24);
  };

  // prologue: rows y0-3 .. y0+3
  for (int v = lane; v < 70; v += 64) {
    const int r = v / 10, c = v - 10 * r;
    park(it.y0 - 3 + r, c, *(const u32x4 *)gaddr(it.y0 - 3 + r, c));
  }
  u32x4 pf[2];
#pragma unroll
  for (int k = 0; k < 2; k++)
    if (son[k]) pf[k] = *(const u32x4 *)gaddr(min(it.y0 + 4 + srow[k], it.h - 1), scol[k]);
  const int g = lane & 31, rr = lane >> 5;
  const int x = it.cx0 + 4 * g;
  const bool colok = x < it.cx1;
  for (int yc = it.y0; yc < it.y1; yc += 8) {
    // entries whose rows the next 8 staged rows would overwrite are classified now (rare in textured areas)
    if (tail_g != head_g && oldest_row(qg, head_g, QG - 1) < yc - 17) pop_groups(true);
    if (tail_f != head_f && oldest_row(qf, head_f, QF - 1) < yc - 17) pop_fast(true);
    if (tail_c != head_c && oldest_row(qc, head_c, QC - 1) < yc - 17) pop_corners(true);
    // rows yc+4 .. yc+11 -> ring; the next chunk's rows -> registers
#pragma unroll
    for (int k = 0; k < 2; k++)
      if (son[k]) park(yc + 4 + srow[k], scol[k], pf[k]);
#pragma unroll
    for (int k = 0; k < 2; k++)
      if (son[k]) pf[k] = *(const u32x4 *)gaddr(min(yc + 12 + srow[k], it.h - 1), scol[k]);
    lds_wait();
#pragma unroll 1
    for (int i = 0; i < 4; i++) {
      const int y = yc + 2 * i + rr;
      const lds_u8 *pc = rowptr(y) + x;
      const uint32_t wc = *(const lds_u32 *)pc;
      const uint32_t wl = *(const lds_u32 *)(pc - 4);
      const uint32_t wr = *(const lds_u32 *)(pc + 4);
      const uint32_t wu = *(const lds_u32 *)(pc - 3 * P);
      const uint32_t wd = *(const lds_u32 *)(pc + 3 * P);
      const uint32_t sv = max(__builtin_amdgcn_sad_u8(wu, wc, 0u), __builtin_amdgcn_sad_u8(wd, wc, 0u));
      const uint32_t sh = max(__builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(wc, wl, 1), wc, 0u),
                              __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(wr, wc, 3), wc, 0u));
      const bool pass = colok && y < it.y1 && min(sv, sh) > (uint32_t)THR;
      const uint64_t m = __ballot(pass);
      if (m == 0) continue;
      if (pass) qg[(tail_g + pdev::ballot_rank(m)) & (QG - 1)] = (uint32_t)x | ((uint32_t)y << 12);
      tail_g += (uint32_t)__popcll(m);
      pop_groups(false);
    }
    asm volatile("" : "+v"(pf[0]), "+v"(pf[1]));     // keep the prefetch loads issued above the chunk's work
  }
  pop_groups(true);
  pop_fast(true);
  pop_corners(true);
  // reduce the checksum over the wave
  for (int o = 32; o > 0; o >>= 1) csum += (uint32_t)__shfl_xor((int)csum, o, 64);
  if (lane == 0) {
    out_count[(size_t)blockIdx.y * nitems + blockIdx.x] = ncorner + (nscored << 16);
    out_sum[(size_t)blockIdx.y * nitems + blockIdx.x] = nbatch_h | (nbatch_f << 10) | (nbatch_p << 21);   // (batch counts)
  }
}

// straightforward reference on the GPU from the same validated primitives: one thread per pixel of pyramid 0
__global__ void k_reference(const uint8_t *__restrict__ pyr, uint8_t *__restrict__ score_map, int row0, int w, int h, int xend) {
  const int x = B + blockIdx.x * blockDim.x + threadIdx.x, y = B + blockIdx.y;
  if (x >= xend || y >= h - B) return;
  const uint8_t *c = pyr + (size_t)(row0 + y) * VSTEP + x;
  if (!pdev::fast9(c, VSTEP, THR)) return;
  const uint8_t sc = x < w - B ? pdev::harris_score(c, VSTEP, HTHR) : (uint8_t)0xff;
  score_map[(size_t)(row0 + y) * VSTEP + x] = sc ? sc : (uint8_t)1;
}

// ---- host ------------------------------------------------------------------------------------------------------
static bool fast9_cpu(const uint8_t *img, int x, int y) {
  static const int dy[16] = {-3, -3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2};
  static const int dx[16] = {-1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2};
  const int c = img[y * VSTEP + x];
  unsigned bm = 0, dm = 0;
  for (int k = 0; k < 16; k++) {
    const int p = img[(y + dy[k]) * VSTEP + x + dx[k]];
    if (p > c + THR) bm |= 1u << k;
    if (p < c - THR) dm |= 1u << k;
  }
  auto arc = [](unsigned m) {
    m |= m << 16;
    for (int s = 0; s < 16; s++)
      if (((m >> s) & 0x1ff) == 0x1ff) return true;
    return false;
  };
  return arc(bm) || arc(dm);
}

int main(int argc, char **argv) {
  if (argc < 2) return 1;
  const int batch = argc > 2 ? atoi(argv[2]) : 256, seg = argc > 3 ? atoi(argv[3]) : 112;
  static const int LW[8] = {640, 533, 444, 370, 309, 257, 214, 179}, LH[8] = {480, 400, 333, 278, 231, 193, 161, 134};
  std::vector<Item> items;
  int row0 = 0;
  for (int l = 0; l < 8; l++) {
    const int nx = LW[l] - 2 * B, xend = B + 16 * ((nx + 15) / 16);
    const int ny = LH[l] - 2 * B, nseg = (ny + seg - 1) / seg, sh = (((ny + nseg - 1) / nseg) + 7) & ~7;
    for (int cx0 = B; cx0 < xend; cx0 += 128)
      for (int y0 = B; y0 < LH[l] - B; y0 += sh) items.push_back({row0, LH[l], cx0, std::min(cx0 + 128, xend), y0, std::min(y0 + sh, LH[l] - B), LW[l] - B});
    row0 += LH[l];
  }
  const size_t pyr = (size_t)ROWS * VSTEP;
  std::vector<uint8_t> h(pyr * batch);
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 2;
  const size_t got = fread(h.data(), 1, h.size(), f) / pyr;
  fclose(f);
  if (got == 0) return 3;
  for (size_t b = got; b < (size_t)batch; b++) memcpy(&h[b * pyr], &h[(b % got) * pyr], pyr);
  uint8_t *d_pyr; Item *d_items; uint32_t *d_cnt, *d_sum;
  hipMalloc(&d_pyr, h.size()); hipMalloc(&d_items, items.size() * sizeof(Item));
  hipMalloc(&d_cnt, items.size() * batch * 4); hipMalloc(&d_sum, items.size() * batch * 4);
  hipMemcpy(d_pyr, h.data(), h.size(), hipMemcpyHostToDevice);
  hipMemcpy(d_items, items.data(), items.size() * sizeof(Item), hipMemcpyHostToDevice);
  const dim3 grid((unsigned)items.size(), (unsigned)batch);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k_bands, grid, dim3(64), LDS_BYTES, 0, d_items, d_pyr, d_cnt, d_sum, (int)items.size(), (uint8_t *)nullptr);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 50;
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k_bands, grid, dim3(64), LDS_BYTES, 0, d_items, d_pyr, d_cnt, d_sum, (int)items.size(), (uint8_t *)nullptr);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // score maps of pyramid 0: band kernel vs the per-pixel reference kernel
  uint8_t *d_map, *d_ref;
  hipMalloc(&d_map, pyr); hipMalloc(&d_ref, pyr); hipMemset(d_map, 0, pyr); hipMemset(d_ref, 0, pyr);
  hipLaunchKernelGGL(k_bands, dim3((unsigned)items.size(), 1), dim3(64), LDS_BYTES, 0, d_items, d_pyr, d_cnt, d_sum, (int)items.size(), d_map);
  {
    int r0 = 0;
    for (int l = 0; l < 8; l++) {
      const int nx = LW[l] - 2 * B, xend = B + 16 * ((nx + 15) / 16);
      hipLaunchKernelGGL(k_reference, dim3((xend - B + 63) / 64, LH[l] - 2 * B), dim3(64), 0, 0, d_pyr, d_ref, r0, LW[l], LH[l], xend);
      r0 += LH[l];
    }
  }
  hipDeviceSynchronize();
  std::vector<uint8_t> hmap(pyr), href(pyr);
  hipMemcpy(hmap.data(), d_map, pyr, hipMemcpyDeviceToHost);
  hipMemcpy(href.data(), d_ref, pyr, hipMemcpyDeviceToHost);
  size_t map_diff = 0, map_nz = 0;
  for (size_t i = 0; i < pyr; i++) {
    map_diff += hmap[i] != href[i];
    map_nz += href[i] > 1;
  }
  hipLaunchKernelGGL(k_bands, grid, dim3(64), LDS_BYTES, 0, d_items, d_pyr, d_cnt, d_sum, (int)items.size(), (uint8_t *)nullptr);
  hipDeviceSynchronize();
  std::vector<uint32_t> cnt(items.size() * batch);
  hipMemcpy(cnt.data(), d_cnt, cnt.size() * 4, hipMemcpyDeviceToHost);
  unsigned long long total = 0;
  std::vector<uint32_t> bsum(items.size() * batch);
  hipMemcpy(bsum.data(), d_sum, bsum.size() * 4, hipMemcpyDeviceToHost);
  unsigned long long nh = 0, nf = 0, np = 0;
  for (uint32_t v : bsum) { nh += v & 1023; nf += (v >> 10) & 2047; np += v >> 21; }
  printf("batches per launch: pretest %llu, FAST %llu, Harris %llu\n", np, nf, nh);
  unsigned long long scored = 0;
  for (uint32_t &c : cnt) { scored += c >> 16; c &= 0xffff; total += c; }
  // CPU check on pyramid 0
  size_t bad = 0; unsigned long long ref_total = 0;
  for (size_t i = 0; i < items.size(); i++) {
    const Item &it = items[i];
    uint32_t n = 0;
    for (int y = it.y0; y < it.y1; y++)
      for (int x = it.cx0; x < it.cx1; x++) n += fast9_cpu(h.data() + (size_t)it.row0 * VSTEP, x, y);
    ref_total += n;
    if (n != cnt[i]) {
      if (bad < 5) printf("item %zu (row0 %d cx %d..%d y %d..%d): gpu %u cpu %u\n", i, it.row0, it.cx0, it.cx1, it.y0, it.y1, cnt[i], n);
      bad++;
    }
  }
  printf("band probe: %zu items per pyramid (segments of <= %d rows), LDS %d B per wave, %.4f ms per launch of %d pyramids, "
         "%llu FAST corners (%.0f per pyramid), %.0f non-zero Harris scores per pyramid; pyramid 0: %zu of %zu items differ from "
         "the CPU FAST-9 (cpu total %llu), score map vs the per-pixel reference kernel: %zu bytes differ (%zu scores)\n",
         items.size(), seg, LDS_BYTES, ms / reps, batch, total, (double)total / batch, (double)scored / batch, bad, items.size(), ref_total,
         map_diff, map_nz);
  return bad || map_diff ? 4 : 0;
}
