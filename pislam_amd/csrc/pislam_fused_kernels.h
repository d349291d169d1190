// pislam_fused_kernels.h — the measured path: fused detect + score + NMS per strip, then gather + ORB.
//
// A strip = R rows x the full width of one level of one pyramid; a 256-thread workgroup walks a run of
// consecutive strips of one level (k_fused_strips).  The image strip (+halo) is staged into LDS with
// 16-byte coalesced loads; everything up to the keypoint list happens out of LDS, so the image is read
// from HBM once and the reference's score map (`out`, Fast.h:54 / Fast.h:166) is never materialised
// in HBM:
//
//   phase A0 SAD prefilter on every 4-pixel group (necessary condition of the compass test): the classified
//            rows are scanned as one linear run of tile bytes, 256 per wave step
//   phase A1 "two adjacent compass points" test on the surviving groups, 4 pixels per lane in the byte
//            domain (on halved differences: a necessary condition, one grey level wider than the exact test
//            on one side); survivors are compacted per wave (ballot + mbcnt) into a wave-private LDS queue
//   phase B  whenever a queue holds >= 64 entries the wave pops 64 and runs the full FAST-9 arc test
//            on densely packed lanes (result-identical to Fast.h:63-147); corners go to ONE queue per
//            workgroup.  (Candidates travel through A0 / A1 / B as tile byte offsets.)
//   phase C  all waves score that queue: 6x6 Harris (Harris.h:80-248); entries become x | row << 16 | score << 24
//   phase D  2x2-block NMS (Fast.h:228-312), driven by the queue of non-zero scores; survivors are
//            ranked into block-raster order and written to the strip's slots of a staging buffer
//
// Two LDS layouts (strip_lds): ALIAS (product: the score tile is laid over the dead image rows after
// phase C, scores travel in the queue entries; strips whose queues overflow go to an overflow list) and
// plain (separate score tile, scan fallbacks: k_fused_overflow and the option alias=0).
// k_gather_orb turns per-strip counts into offsets (exclusive scan in strip order = reference order),
// copies the staged keypoints to their final positions and describes them (orbCompute, Orb.h:396).
// Deterministic: no result depends on the order of atomics.
#pragma once
#include "pislam_dev.h"

// development switches (tools/ab_build.sh -D...): the defaults are the product
#ifndef PISLAM_FRAME_SLEEP
#define PISLAM_FRAME_SLEEP 16 // k_frame: s_sleep argument between polls (x 64 cycles)
#endif
#ifndef PISLAM_OVL
#define PISLAM_OVL 1          // strips of a run follow each other without a workgroup barrier (strip_body)
#endif
#ifndef PISLAM_DIAG
#define PISLAM_DIAG 1         // pretest also requires two adjacent DIAGONAL ring points (3/7/11/15) of one polarity
#endif
#ifndef PISLAM_ORB_PITCH_DW
#define PISLAM_ORB_PITCH_DW 12   // dwords per ORB patch row in LDS (12 = round 1-5: 48-byte rows skewed by a dword per 8 rows)
#endif

namespace pf {

using namespace pdev;

// Explicit LDS (address space 3) pointer types: with queues, lambdas and selects in play the
// compiler's address-space inference gives up and falls back to FLAT loads, which are several
// times slower than ds_read.  Typed pointers keep every tile / queue access a DS instruction.
typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // plain clang vector (uint4 is a class)
typedef __attribute__((address_space(3))) u32x4 lds_u4;

constexpr int MAX_ORDER = 160;              // runs per pyramid the launch-order table holds
constexpr int MAX_LEVELS = 24;              // plan entries: levels, wide ones cut into x-tiles (see FusedLevel::gn)
constexpr int WAVES = 4;               // waves per strip workgroup
constexpr int NT = WAVES * 64;         // threads per strip workgroup
constexpr int QCAP_G = 128;            // 4-pixel groups that passed the SAD prefilter (< 64 before a <= 64 push)
constexpr int QCAP_F = 192;            // FAST candidates (< 64 before a <= 128 half-push)
constexpr int QCAP = QCAP_G + QCAP_F;  // dwords of private queue space per wave
constexpr int QH_SHARED = 1024;        // workgroup-shared queue of corners awaiting their Harris score (a 28-row
                                       // strip of a textured photo holds up to ~900: the reference's demo image)
constexpr int QN_SHARED = 512;         // plain layout only: queue of pixels with a non-zero score (NMS candidates)
constexpr int QS_SHARED = 256;         // survivors of one strip awaiting their rank
constexpr int SHARED_Q = QH_SHARED + QN_SHARED;
constexpr int NMS_SCRATCH = 3 * QS_SHARED;   // dwords: survivors, their keys (then ranks), bucket keep flags (in the idle per-wave queues)
static_assert(NMS_SCRATCH <= WAVES * QCAP, "NMS scratch must fit the per-wave queue area");
static_assert(QS_SHARED <= NT, "one survivor per thread (ranks are held in a register across a barrier)");
constexpr uint32_t STRIP_DESCRIBED = 0x80000000u;   // strip_count bit 31: the strip wrote its keypoints' descriptors

struct FusedLevel {
  int w, h;          // level size
  int row0, col0;    // position in the stacked pyramid
  int R;             // strip height (even)
  int nstrips;       // strips in this level
  int strip0;        // index of the level's first strip within the pyramid
  int slot0;         // staging slot (in keypoints) of the level's first strip
  int nbx;           // 2x2 blocks per block-row
  int xend;          // one past the last classified column (Fast.h:61,149)
  int pitch;         // LDS SCORE tile pitch in bytes (multiple of 16): the score tile spans the full level width
  int tpitch;        // LDS IMAGE tile pitch in bytes (multiple of 16) = classified columns + halo
  int nruns, run0;   // runs (= workgroups) of this level: run_len consecutive strips each
  int apad;          // ALIAS layout: bytes between the per-wave queues and the image tile (multiple of 16)
  int qh;            // ALIAS layout: entries of the shared corner queue (>= QH_SHARED: the plan hands the LDS
                     // a level's narrower tiles leave under the residency budget to its queue)
  int tbytes;        // plain layout: bytes reserved for the image tile = max((R+10)*tpitch, the scan
                     // fallbacks' survivor / per-cell buffers — larger only with narrow x-tiles)
  uint32_t vpr_recip; // ceil(2^32 / (tpitch/16)): row = umulhi(i, vpr_recip) for a vector index i < 2^16
  uint32_t tp_recip;  // ceil(2^32 / tpitch): tile row of a byte offset k < 2^16 into the tile = umulhi(k, tp_recip)
  // A plan entry is a whole pyramid level, or one X-TILE of a wide level: a column range handled by its own
  // workgroups like a level of its own (w / col0 describe the tile plus a halo of real image columns in place
  // of the border), so that the LDS footprint — and with it the number of resident workgroups — does not grow
  // with the level width.  Tiles classify and score a few columns beyond the blocks they own (the NMS of a
  // boundary block reads its neighbours' scores), emit only the blocks whose origin lies in [ex0, ex1), and
  // k_gather_orb merges the tiles' per-strip lists back into the level's block-raster order.
  int ex0, ex1;      // block origins (entry-relative x) this entry emits: [ex0, ex1)   (whole level: [B, w-B))
  int xscore;        // entry-relative x of the level's w-B: columns at / beyond it keep 0xff (Fast.h:172)
  int gfirst, gn;    // the entries gfirst .. gfirst+gn-1 are the tiles of this entry's level (same R, nstrips)
};

struct FusedParams {
  int nlevels, strips_per_pyr, slots_per_pyr;
  int runs_per_pyr, run_len;   // a workgroup walks run_len consecutive strips of one level (see k_fused_strips)
  int vstep, rows, border, thr;
  int32_t hthr;
  int batch;
  int lbs, limit;    // fastExtract logBucketSize (0 = none; fused path: 2..5) and bucketLimit
  int words;         // orbCompute words (descriptor dwords per keypoint)
  int orb_in_strip;  // ALIAS + 16-byte-aligned kernels: strips describe their own keypoints (strip_body phase E)
  int order_n;       // > 0: the launch order of a pyramid's runs is order[] (longest first)
  int dump_score;    // debug: also write the score tile to the HBM score map
  int ablate;        // profiling only: bit0 stop after staging, bit1 pretest only, bit2 no Harris, bit3 no NMS
  FusedLevel lv[MAX_LEVELS];
  // Runs of one pyramid in launch order, longest (estimated) first, so that the short ones fill the tail of
  // the launch (longest-processing-time-first list scheduling); plans with more runs keep entry order.
  uint32_t order[MAX_ORDER];                       // entry << 16 | run inside the entry (dwords: scalar loads)
};

// Workgroup barrier that orders LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier).  __syncthreads()
// also waits for every outstanding GLOBAL load (vmcnt(0)), which would serialise the next strip's
// prefetch with the first barrier of the current strip.  Nothing in the strip kernel communicates
// through global memory inside a workgroup.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ONE lane's LDS atomic add with return as ONE instruction.  A source-level atomicAdd inside `if (lane == 0)` is
// wrapped by hipcc's atomic optimiser in a wave reduction of its own (2 v_mbcnt, compare, multiply-add, two moves:
// 7 VALU for an add that a single lane executes) — once per FAST batch and per NMS batch.
__device__ __forceinline__ uint32_t lds_add_rtn(uint32_t *p, uint32_t v) {
  uint32_t old;
  const uint32_t a = (uint32_t)(uintptr_t)(lds_u32 *)p;
  asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(a), "v"(v) : "memory");
  return old;
}

constexpr int PF_MAX = 4;              // 16-byte vectors a thread can hold for the next strip's prefetch

// candidate coordinates packed as x | (tile_row << 16)
__device__ __forceinline__ uint32_t pack_xy(int x, int r) { return (uint32_t)x | ((uint32_t)r << 16); }

// Necessary condition for a 9-arc: an arc of 9 ring positions contains two ADJACENT compass
// points (ring indices 1,5,9,13), so both must be dark (or both bright): one of the vertical pair (p1, p9) AND one of
// the horizontal pair (p5, p13) on the same side of the centre.  strip_body's pretest_batch evaluates it for the 4
// pixels of a dword at once, in the byte domain.

// Rare path of the plain layout (shared corner queue full): score the flagged lanes right away.  Kept
// out of line so that the hot loops of the kernel do not carry a second inlined copy of the Harris
// arithmetic.  (Returning the score to push it as an NMS candidate instead was measured: the call then
// keeps 11 more VGPRs live in the caller and costs a resident workgroup.)
__device__ __attribute__((noinline)) void harris_overflow(lds_u8 *tile, lds_u8 *sc, int tpitch, int pitch, int32_t hthr,
                                                          bool valid, uint32_t e) {
  if (valid) {
    const int x = e & 0xffff, r = (e >> 16) & 0xff;
    sc[r * pitch + x] = harris_score_mm(tile + r * tpitch + x - 3, tpitch, hthr);
  }
}

// ===========================================================================
// ORB for two keypoints per wave (one per 32-lane half) — orbCompute, Orb.h:396-441 — shared by the strip
// kernel (keypoints described right after their strip's NMS, while the rows they need are still in the L2
// that just served the strip) and by k_gather_orb (keypoints of strips that took a fallback path).
//   lane r of a half owns patch row dy = r-15.  Each row's 48-byte window covering x-15..x+16 is fetched by ONE
//   LANE QUAD as four 12-byte pieces (global_load_dwordx3; round 6 — until then three 16-byte chunks by adjacent
//   lanes: PISLAM_FETCH_X3 0), parked in an LDS patch (pitch 48) and read back patch-aligned with 9 aligned dwords +
//   v_alignbyte; the circle mask (Orb.h:118-121,
//   163-286) is a per-lane constant; the moments are v_dot4_u32_u8 dot products with the |dx| weights
//   (Orb.h:123-126); rows are summed across the 32 lanes by DPP; every lane evaluates the angle bin
//   (Orb.h:310-387); the 256 BRIEF tests (Brief.h:52) read the LDS patch through the precomputed offset
//   table g_brief_ofs, eight tests per lane = one descriptor byte per lane.
// ===========================================================================
constexpr int OWAVES = 4;                           // waves per k_gather_orb workgroup
// One 48-byte window per patch row.  Default layout (PISLAM_ORB_PITCH_DW 12): rows 12 dwords apart, skewed by one dword per
// 8 rows — the 9 row reads of the moments (lane = row) are conflict-free (a pitch of 12 dwords alone puts rows r, r+8, r+16,
// r+24 on the same banks), and so are the three dword stores that park a 12-byte piece (lanes = (row, piece): 12 row + 3
// piece + i hits 32 different banks for the 8 rows x 4 pieces of a half-wave).  With the 16-byte chunks of rounds 2-5 the
// four stores of a chunk fell on 8 banks (4-way: 5.6 M of the kernel's 13.8 M conflict cycles per launch, 1.5 M now); rows 13
// dwords apart (PISLAM_ORB_PITCH_DW 13, round 6) cured that at the price of 1 KB more LDS per workgroup and were not kept.
constexpr int ORB_PITCH = 4 * PISLAM_ORB_PITCH_DW;
__host__ __device__ constexpr int orb_row_ofs(int r) { return PISLAM_ORB_PITCH_DW == 12 ? r * 48 + 4 * (r >> 3) : r * ORB_PITCH; }
// 31 rows (the idle lane's row 31 reads harmless bytes of whatever follows) + slack for the byte shift, a multiple of 16
constexpr int ORB_PATCH_BYTES = PISLAM_ORB_PITCH_DW == 12 ? 32 * 48 + 32 : (31 * ORB_PITCH + 4 + 15) / 16 * 16;

// Sums over each 32-lane half of the wave, results in every lane of that half: two values at once.  DPP row shifts
// (zero fill) leave each 16-lane row's sum in its last lane, row_bcast:15 folds row 0 into row 1 and
// row 2 into row 3, two v_readlane per value pick the totals up.  The two chains' steps alternate: a DPP step must
// wait two cycles for its own previous result, the other chain's step fills one of them.
__device__ __forceinline__ void half_sum2(int &a, int &b, int half) {
  a += __builtin_amdgcn_update_dpp(0, a, 0x111, 0xf, 0xf, true);
  b += __builtin_amdgcn_update_dpp(0, b, 0x111, 0xf, 0xf, true);
  a += __builtin_amdgcn_update_dpp(0, a, 0x112, 0xf, 0xf, true);
  b += __builtin_amdgcn_update_dpp(0, b, 0x112, 0xf, 0xf, true);
  a += __builtin_amdgcn_update_dpp(0, a, 0x114, 0xf, 0xf, true);
  b += __builtin_amdgcn_update_dpp(0, b, 0x114, 0xf, 0xf, true);
  a += __builtin_amdgcn_update_dpp(0, a, 0x118, 0xf, 0xf, true);
  b += __builtin_amdgcn_update_dpp(0, b, 0x118, 0xf, 0xf, true);
  a += __builtin_amdgcn_update_dpp(0, a, 0x142, 0xa, 0xf, false);
  b += __builtin_amdgcn_update_dpp(0, b, 0x142, 0xa, 0xf, false);
  const int a0 = __builtin_amdgcn_readlane(a, 31), a1 = __builtin_amdgcn_readlane(a, 63);
  const int b0 = __builtin_amdgcn_readlane(b, 31), b1 = __builtin_amdgcn_readlane(b, 63);
  a = half ? a1 : a0;
  b = half ? b1 : b0;
}

#ifndef PISLAM_FETCH_X3
#define PISLAM_FETCH_X3 1
#endif
#if PISLAM_FETCH_X3
// One patch row per lane QUAD: the row's 48-byte window as four 12-byte pieces in four ADJACENT lanes (global_load_dwordx3), so
// that the four lanes the texture path takes up together fall into one cache line (two when the window straddles a line)
// instead of the 2.6 lines of "three 16-byte chunks per row" — and the three dword stores that park a piece (lanes 12 bytes
// apart, rows 48 apart) spread over the banks where the four stores of a 16-byte chunk fell on 8.  Measured per launch (256 VGA
// pyramids): vector-L1 tag look-ups 22.4 -> 18.8 M, line requests to the L2 unchanged (6.1 M: lines are shared between the two
// keypoints of one load instruction only), LDS bank-conflict cycles 13.8 -> 9.8 M, the kernel 1.4 us shorter.
constexpr int ORB_NLD = 4;                          // loads per lane and pair: 62 quads = 248 of 256 slots
struct OrbPiece {
  uint32_t x, y, z;
};
struct OrbWin {
  OrbPiece w[ORB_NLD];
};
#else
constexpr int ORB_NLD = 3;
struct OrbWin {
  uint4 w[3];
};
#endif
// Per-lane geometry of the pair scheme: slot j = lane + 64 j -> (keypoint half, patch row, 16-byte chunk) with
// the three chunks of a row in ADJACENT lanes (the L1 processes a divergent load at about one distinct cache
// line per cycle: one or two lines per row instead of three requests).
struct OrbLane {
  int half, r;                                      // r = patch row index, dy = r - 15 (r = 31 idle)
  int sl_h[ORB_NLD], sl_park[ORB_NLD], sl_rel[ORB_NLD];
  bool sl_on[ORB_NLD];
  // circle mask of this lane's row (Orb.h:118-121,163-286), folded into dot-product weights — byte j of the 32
  // covers dx = j - 15: m01 = 1 where the pixel belongs to the patch, mdx = |dx| there (0 elsewhere)
  uint32_t m01[8], mdx[8];
};
__device__ __forceinline__ OrbLane orb_lane(int lane, int vstep) {
  OrbLane G;
  G.half = lane >> 5;
  G.r = lane & 31;
  // (the row's 16 mask words come from a compile-time table: four 16-byte loads instead of ~150 VALU)
  const uint32_t *row = ::g_orb_masks.v[G.r];        // defined by the including TU
#pragma unroll
  for (int k = 0; k < 8; k++) {
    G.m01[k] = row[k];
    G.mdx[k] = row[8 + k];
  }
#if PISLAM_FETCH_X3
#pragma unroll
  for (int j = 0; j < ORB_NLD; j++) {
    const int slot = lane + 64 * j;                 // 0..255, 248 used: quad q = slot / 4 -> (keypoint half, patch row)
    const int q = slot >> 2, k = slot & 3;
    const int h = q >= 31 ? 1 : 0, row = q - 31 * h;
    G.sl_h[j] = h;
    G.sl_on[j] = q < 62;
    G.sl_rel[j] = row * vstep + 12 * k;             // byte offset of the piece relative to the (16-byte aligned) window origin
    G.sl_park[j] = h * ORB_PATCH_BYTES + orb_row_ofs(row) + 12 * k;
  }
#else
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const int slot = lane + 64 * j;                 // 0..191, 186 used
    const int h = slot >= 93 ? 1 : 0, within = slot - 93 * h;
    const int row = within / 3, chunk = within - 3 * row;
    G.sl_h[j] = h;
    G.sl_on[j] = slot < 186;
    G.sl_rel[j] = row * vstep + 16 * chunk;         // byte offset of the chunk relative to the patch origin (y-15, x-15)
    G.sl_park[j] = h * ORB_PATCH_BYTES + orb_row_ofs(row) + 16 * chunk;
  }
#endif
  return G;
}
// Issue the loads of one pair (p0 / p1: packed keypoints in stacked coordinates, 0 = absent).  vstep % 16 == 0
// makes the byte shift row-independent.  A chunk that starts outside the pyramid (absent keypoint, idle slot,
// window slack past the last row) holds no byte any patch uses — the buffer size is a multiple of 16 — so it
// is simply clamped into the buffer instead of being zero-filled.  32-bit byte offsets: a pyramid is < 2 GiB;
// a negative offset wraps to a huge unsigned value and is clamped too.
__device__ __forceinline__ OrbWin orb_fetch(const OrbLane &G, uint32_t p0, uint32_t p1, const uint8_t *__restrict__ im,
                                            int vstep, uint32_t img_bytes32) {
  OrbWin f;
  const int org0 = (decode_y(p0) - 15) * vstep + (decode_x(p0) - 15);
  const int org1 = (decode_y(p1) - 15) * vstep + (decode_x(p1) - 15);
  const int d01 = org1 - org0;                      // (wave-uniform: the slot's keypoint is picked by one v_and, not by a select)
#if PISLAM_FETCH_X3
#pragma unroll
  for (int j = 0; j < ORB_NLD; j++) {
    // (vstep % 16 == 0: aligning the patch origin aligns every row's window; a piece is 4-byte aligned.  A piece that starts
    //  within 12 bytes of the end of the buffer is moved back as a whole — its bytes are then NOT where the patch expects
    //  them: harmless, because such a piece holds no patch byte.  The batch entry points require border >= 16, so the last row
    //  a patch uses is at most the buffer's last row but one, a piece reaches at most 17 bytes past a patch byte, and a
    //  pyramid row is at least that long: every piece that holds a patch byte ends inside the buffer.)
    const uint32_t a = min(((uint32_t)(org0 + (d01 & -G.sl_h[j])) & ~15u) + (uint32_t)G.sl_rel[j], img_bytes32 - 12u);
    f.w[j] = *(const OrbPiece *)(im + a);           // three adjacent dwords at 4-byte alignment: ONE global_load_dwordx3
  }
#else
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const uint32_t a = min((uint32_t)(org0 + G.sl_rel[j] + (d01 & -G.sl_h[j])) & ~15u, img_bytes32 - 16u);
    f.w[j] = *(const uint4 *)(im + a);
  }
#endif
  return f;
}
// Moments -> angle bin -> BRIEF -> descriptor store of a pair whose patches lie in LDS: `prow4` = this lane's patch row, rounded
// down to a dword (the row's first byte is prow4 + sh3), `bp` = the half's patch origin (byte (dy, dx) at bp + tab offset of
// (dy + 15, dx + 15)), `tab` = the BRIEF offset table of that LDS layout ([rot][t][r], see make_brief_ofs).  (The tail of orb_describe, kept apart
// from the parking so that other LDS layouts can share it: round 6 measured one — row bands staged once per strip,
// tools/probes/band_orb_experiment.patch, docs/experiments.md R6.)
template <class TAB, bool GHOOKS = false>
__device__ __forceinline__ void orb_describe_rows(const OrbLane &G, const lds_u8 *prow4, const uint32_t sh3, const lds_u8 *bp,
                                                  const uint32_t *__restrict__ tab_base, const bool valid, const TAB *rtab,
                                                  int words, uint8_t *__restrict__ dst_base, const uint32_t dst_word,
                                                  const int glevel = 0) {
  const int half = G.half, r = G.r;
  // read the row back aligned to the PATCH: 9 aligned dwords + v_alignbyte (byte-unaligned
  // ds_read_b32 works on gfx950 but costs ~47 stall cycles each — SQ_LDS_UNALIGNED_STALL)
  uint32_t row[8];
  {
    uint32_t in[9];
#pragma unroll
    for (int k = 0; k < 9; k++) in[k] = *(const lds_u32 *)(prow4 + 4 * k);
#pragma unroll
    for (int k = 0; k < 8; k++) row[k] = __builtin_amdgcn_alignbyte(in[k + 1], in[k], sh3);
  }
  // moments of this row: sum v and sum |dx| v, left (dx<0) and right (dx>0) separately (Orb.h:123-126).  The
  // circle mask lives in the dot-product weights (a zero weight ignores the pixel), so the row is never ANDed.
  uint32_t sv = 0, left = 0, right = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) sv = __builtin_amdgcn_udot4(row[k], G.m01[k], sv, false);
#pragma unroll
  for (int k = 0; k < 4; k++) left = __builtin_amdgcn_udot4(row[k], G.mdx[k], left, false);        // dx -15 .. 0
#pragma unroll
  for (int k = 4; k < 8; k++) right = __builtin_amdgcn_udot4(row[k], G.mdx[k], right, false);      // dx 1 .. 15 (16 masked)
  // (the two reductions interleaved: each DPP step must wait two cycles for its own previous result — the other chain's step
  //  fills one of them; |dy| <= 16, sv < 2^13: the full-rate 24-bit multiply)
  int m10 = (int)right - (int)left, m01 = __mul24(r - 15, (int)sv);
  half_sum2(m10, m01, half);
  if (GHOOKS && glevel == 4) {
    asm volatile("" ::"v"(m10), "v"(m01));
    return;
  }
  const uint32_t rot = angle_bin_fast(m10, m01, rtab);
  if (GHOOKS && glevel == 5) {
    asm volatile("" ::"v"(rot));
    return;
  }
  // BRIEF: lane r of a half runs the EIGHT tests k = 8 r .. 8 r + 7 of its keypoint — byte r of the descriptor (bit
  // k % 32 of word k / 32 = byte k / 8, bit k % 8).  The sign of a - b IS the test (Brief.h:52) and v_alignbit_b32
  // shifts it into the byte (tests taken from t = 7 down, so that test 8 r + t ends up in bit t); four adjacent lanes'
  // bytes make a word (two DPP quad permutes + two v_perm_b32) and ds_bpermute_b32 brings the eight words of a half
  // into its lanes 0 .. 7, which store them.  g_brief_ofs is laid out for this: entry [rot][t][r] = test 8 r + t, so
  // that the lanes of a half read 128 contiguous bytes per t.
  // (One test per lane and round, a ballot per round and v_writelane_b32 to bring the 16 half-ballots back into lanes
  //  took 8 + 16 VALU and a 5-cycle hazard pad for the same bits: 24.05 -> 22.9 M VALU per launch together with the
  //  32-bit destination offsets below.  The first measurements of this form were 1.3 % SLOWER per pipelined step: it
  //  needs fewer SGPRs, which let a seventh wave per SIMD in — see the occupancy pin in k_gather_orb.)
  const uint32_t *tab = tab_base + rot * 256 + r;
  // the patch's byte (dy,dx) sits at orb_row_ofs(dy+15) + sh + dx+15 (the table holds the first and last term); sh is
  // the same for every row
  uint32_t ent[8];
#pragma unroll
  for (int t = 0; t < 8; t++) ent[t] = tab[32 * t];        // all 8 table loads in flight
  if (GHOOKS && glevel == 6) {
#pragma unroll
    for (int t = 0; t < 8; t++) asm volatile("" ::"v"(ent[t]));
    return;
  }
  uint32_t pa[8], pb[8];
#pragma unroll
  for (int t = 0; t < 8; t++) pa[t] = bp[ent[t] & 0xffffu], pb[t] = bp[ent[t] >> 16];   // all 16 reads in flight
  uint32_t acc = 0;
#pragma unroll
  for (int t = 7; t >= 0; t--) acc = __builtin_amdgcn_alignbit(acc, pa[t] - pb[t], 31);   // acc = acc << 1 | [a < b]
  const uint32_t n1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0xf5, 0xf, 0xf, true);     // quad_perm [1,1,3,3]
  const uint32_t hw = __builtin_amdgcn_perm(n1, acc, 0x0c0c0400u);                                   // byte r | byte r+1 << 8
  const uint32_t n2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hw, 0xaa, 0xf, 0xf, true);      // quad_perm [2,2,2,2]
  const uint32_t word4 = __builtin_amdgcn_perm(n2, hw, 0x05040100u);                                 // valid in lanes r % 4 == 0
  const uint32_t word = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(4u * (32u * (uint32_t)half + 4u * ((uint32_t)r & 7u))), (int)word4);
  if (GHOOKS && glevel == 7) {
    asm volatile("" ::"v"(word));
    return;
  }
  // (32-bit byte offset from a wave-uniform base: one multiply and a shift-add instead of a 64-bit multiply-add chain)
  if (valid && r < words) *(uint32_t *)(dst_base + (dst_word * 4u + 4u * (uint32_t)r)) = word;
}

// Park the fetched windows of a pair, then moments -> angle bin -> BRIEF.  `dst_base` (wave-uniform) + 4 * `dst_word`
// (< 2^30): where this half's keypoint keeps its `words` descriptor words (ignored when that keypoint is absent).  `wave_patches`: 2 x
// ORB_PATCH_BYTES of LDS private to the wave.  `rtab`: the vrecpe estimate table in LDS.
// `glevel` (profiling instantiation of k_gather_orb only, option "ablate" bits 20..23; 0 in every product kernel): stop the
// describe after 2 = the fetch, 3 = + parking the windows, 4 = + read-back and moments, 5 = + angle bin, 6 = + BRIEF offset
// table loads, 7 = + BRIEF sample reads and bit assembly (everything but the descriptor store) — cumulative, so that the
// difference of two PMC passes is what the phase between them costs (tools/pmc_ablate_gather.sh).
template <class TAB, bool GHOOKS = false>
__device__ __forceinline__ void orb_describe(const OrbLane &G, const OrbWin &cur, uint32_t p0, uint32_t p1,
                                             lds_u8 *wave_patches, int vstep, const TAB *rtab, int words,
                                             uint8_t *__restrict__ dst_base, const uint32_t dst_word, const int glevel = 0) {
  const int half = G.half, r = G.r;
  if (GHOOKS && glevel == 2) {                        // the loads must stay: their registers are "used"
#pragma unroll
    for (int j = 0; j < ORB_NLD; j++) asm volatile("" ::"v"(cur.w[j].x), "v"(cur.w[j].y), "v"(cur.w[j].z));
    return;
  }
  const uint32_t pme = half ? p1 : p0;
  const bool valid = pme != 0;
  const int x = decode_x(pme), y = decode_y(pme);
  // byte shift of the patch inside its 16-byte chunks: ((y - 15) * vstep + (x - 15)) & 15 with vstep % 16 == 0
  // (the precondition of this scheme) — the same for every row, and independent of y
  const uint32_t sh = (uint32_t)(x + 1) & 15u;
  // (the skewed row starts are only 4-byte aligned: four dword stores per chunk instead of one 16-byte store)
#pragma unroll
  for (int j = 0; j < ORB_NLD; j++)
    if (G.sl_on[j]) {
      lds_u32 *d = (lds_u32 *)(wave_patches + G.sl_park[j]);
      d[0] = cur.w[j].x;
      d[1] = cur.w[j].y;
      d[2] = cur.w[j].z;
#if !PISLAM_FETCH_X3
      d[3] = cur.w[j].w;
#endif
    }
  if (GHOOKS && glevel == 3) return;
  lds_u8 *patch_l = wave_patches + half * ORB_PATCH_BYTES;
  orb_describe_rows<TAB, GHOOKS>(G, patch_l + orb_row_ofs(r) + (sh & ~3u), sh & 3u, patch_l + sh, ::g_brief_ofs.v, valid, rtab, words,
                                 dst_base, dst_word, glevel);
}

// Scalar copies of the kernel arguments the strip body needs (the by-value FusedParams must not be
// captured by reference anywhere: hipcc then spills the whole 800-byte struct to scratch).
struct StripArgs {
  int border, thr, ablate, dump_score, lbs, limit, vstep, slots_per_pyr, strips_per_pyr;
  int32_t hthr;
  int words, orb;
};

// All phases of one strip of one plan entry (a level, or an x-tile of a wide level — FusedLevel): stage,
// prefilter / pretest / FAST (scores of over-classified columns and corner queue), Harris for the corners,
// NMS on the score tile.  (Staging a strip's image in several x-tiles INSIDE the workgroup — round 1's
// option "xtile_cols" — was measured slower than full-width tiles and is gone: wide levels are cut into
// plan entries run by separate workgroups instead.)
template <bool VEC16, bool HOOKS, bool ALIAS, bool ORBK, bool BUCK>
__device__ __forceinline__ void strip_body(const StripArgs A, const FusedLevel L, const int pyr, const int s,
                                           const int ys, const int ye, lds_u8 *tile0, lds_u8 *sc, lds_u32 *queues,
                                           lds_u32 *shq, uint32_t *sh_ctr, uint32_t *sh_ctr_prev, bool &deferred,
                                           const uint8_t *__restrict__ im, const ptrdiff_t lim,
                                           uint32_t *__restrict__ stage_kp,
                                           uint32_t *__restrict__ strip_count, uint8_t *__restrict__ score_dump,
                                           size_t score_stride, const bool carry, const int tid,
                                           unsigned long long *__restrict__ prof, u32x4 (&pf)[PF_MAX],
                                           const bool pf_have, const bool pf_want, bool &pf_issued,
                                           uint32_t *__restrict__ ovf, const uint32_t ovf_id,
                                           uint32_t *__restrict__ stage_desc, const uint8_t *__restrict__ imb,
                                           const uint32_t img_bytes32) {
  const int B = A.border;
  // BUCK = false: the instantiation knows logBucketSize == 0 (the product kernel of the default mode carries no
  // bucket code)
  const int Albs = BUCK ? A.lbs : 0;
  // Phase E exists in the ORBK instantiations of the ALIAS kernels on 16-byte aligned layouts (vstep % 16 == 0
  // makes a patch's byte shift row-independent; option "orb_in_strip"); elsewhere k_gather_orb describes the
  // keypoints.  (Compiled out of the default kernels: its registers would cost them 10 VGPRs.)
  constexpr bool ORB = ALIAS && VEC16 && ORBK;
  const int pitch = L.pitch, tpitch = L.tpitch;
  const int cxa = B, cxb = L.xend;                  // classified columns [cxa, cxb)
  const int xbase = (cxa - 4) & ~15;                // first staged column (16-byte aligned)
  lds_u8 *tile = tile0 - xbase;                     // tile + row*tpitch + x with entry column x
  // profiling hook (HOOKS kernels only, ablate bit 8192): workgroup wall-clock cycles per phase
  long long t_last = 0;
  if (HOOKS && prof) t_last = clock64();
  auto mark = [&](int phase) {
    if (HOOKS && prof && tid == 0) {
      const long long t = clock64();
      // (a no-return atomic: a read-modify-write would stall wave 0 for a memory round trip at every mark)
      (void)__hip_atomic_fetch_add(&prof[phase], (unsigned long long)(t - t_last), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      t_last = t;
    }
  };
  const int lane = tid & 63;                        // (== lane_id() for the 1-D workgroup)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform (keeps loops scalar)
  // `carry`: the workgroup has just finished the strip above this one (same level, full height R) and
  // its tiles are intact.  The image rows [ys-4, ys+6) are that strip's tile rows R..R+9 and the
  // score rows [ys-1, ys+2) its score rows R..R+2, so they are moved up inside LDS instead of being
  // staged / classified / scored a second time; the NMS candidates queued for those rows move too.
  const bool carry_img = carry;
  // OVL (the product kernels: ALIAS layout, no in-strip buckets, no in-strip ORB): a run's strips follow each other WITHOUT
  // a workgroup barrier in between.  After the NMS barrier of strip s wave 0 ranks and emits its survivors while the
  // other waves already move the halo rows, compact the corner queue (wave 1) and store the prefetched rows of strip
  // s + 1; everybody meets again at the staging barrier.  What makes that safe:
  //  * the counters are double-buffered by strip parity (sh_ctr = this strip's block, sh_ctr_prev = the block of the
  //    strip above): nothing a late wave still reads of strip s is reset before the staging barrier of strip s + 1;
  //  * the carry copy and the store of the prefetched rows touch the same LDS vector from the same THREAD (the copy loop
  //    is dealt like the prefetch: vector v of the new rows belongs to thread v % NT), so the copy's reads of the old
  //    rows R .. R+9 precede the stores over them in one wave's program order — no barrier between copy and store;
  //  * the survivors' ranking scratch lies in front of the image tile and is rewritten only by the next classification.
  constexpr bool OVL = PISLAM_OVL && ALIAS && !BUCK && !ORBK;
  const int cwave = OVL ? 1 : 0;                    // the wave that compacts the carried queue
  bool need_sync = true;                            // a barrier between the carry copy and the staging stores
  if (carry) {
    const int R = L.R;
    if (carry_img) {
      const int nv = (10 * tpitch) >> 4;            // source rows R.. and destination rows 0..9 are disjoint (R >= 10)
      if (OVL && VEC16 && pf_have) {
        // source vector i is new-row vector (R - 10) * vpr + i: copied by the thread that will store over it
        const int i0 = (tid - (R - 10) * (tpitch >> 4)) & (NT - 1);
        for (int i = i0; i < nv; i += NT) ((lds_u4 *)tile0)[i] = ((const lds_u4 *)(tile0 + R * tpitch))[i];
        need_sync = false;
      } else {
        for (int i = tid; i < nv; i += NT) ((lds_u4 *)tile0)[i] = ((const lds_u4 *)(tile0 + R * tpitch))[i];
      }
    }
    if (!ALIAS) {                                   // (ALIAS: the carried scores travel in the queue entries)
      const int nvs = (3 * pitch) >> 4;
      for (int i = tid; i < nvs; i += NT) ((lds_u4 *)sc)[i] = ((const lds_u4 *)(sc + R * pitch))[i];
    }
    if (wave == cwave) {                            // in-place compaction of the candidate queue (front to back)
      // plain: the queue of non-zero scores; ALIAS: the one queue of corners, whose entries carry their
      // score in the top byte (0 = scored below the threshold: dropped here)
      lds_u32 *qn = ALIAS ? shq : shq + QH_SHARED;
      const int tn = (int)sh_ctr_prev[ALIAS ? 0 : 2];
      int kept = 0;
      for (int c0 = 0; c0 < tn; c0 += 64) {
        const uint32_t e = qn[min(c0 + lane, tn - 1)];
        const bool k = c0 + lane < tn && (int)((e >> 16) & 0xff) >= R && (!ALIAS || (e >> 24) != 0);
        const uint64_t m = __ballot(k);
        if (k) qn[kept + ballot_rank(m)] = e - ((uint32_t)R << 16);
        kept += __popcll(m);
      }
      if (lane < 8)
        sh_ctr[lane] = lane == 1 ? (uint32_t)(ALIAS ? L.qh : QH_SHARED) : (lane == 2 || (ALIAS && lane == 0)) ? (uint32_t)kept : 0u;
    }
    if (need_sync) lds_barrier();
    if (!ALIAS) {
      const int nz = (L.R * pitch) >> 4;            // fresh score rows 3 .. R+2
      for (int i = tid; i < nz; i += NT) ((lds_u4 *)(sc + 3 * pitch))[i] = (u32x4)(0u);
    }
  } else {                                          // zero the score tile, reset the counters
    if (!ALIAS) {
      const int nz = ((L.R + 3) * pitch) >> 4;
      for (int i = tid; i < nz; i += NT) ((lds_u4 *)sc)[i] = (u32x4)(0u);
    }
    if (tid < 8) sh_ctr[tid] = tid == 1 ? (uint32_t)(ALIAS ? L.qh : QH_SHARED) : 0u;
  }
  lds_u32 *qg = queues + wave * QCAP;              // 4-pixel groups for the exact pretest
  lds_u32 *qf = qg + QCAP_G;                       // FAST candidates
  // Corners are rare (~1 % of the pixels): a private queue per wave would end in a mostly empty
  // 64-lane Harris batch per wave, so corners go to ONE queue per workgroup (LDS atomic append)
  // and are scored by all waves together once FAST is finished.  (The waves' left-over pretest / FAST
  // candidates, < 64 each, are NOT merged: that would take one more barrier per strip.)
  lds_u32 *shq_h = shq;
  lds_u32 *shq_n = shq_h + QH_SHARED;               // (plain layout only)
  const int qcap = ALIAS ? L.qh : QH_SHARED;         // capacity of the shared corner queue
  int ng = 0, nf = 0;                               // wave-uniform queue fills
  // NOTE: the lambdas below capture by reference; they must only touch LOCAL copies of kernel
  // arguments — capturing `P` itself makes the compiler spill the whole 800-byte struct to scratch.
  const int thr = A.thr;
  const int32_t hthr = A.hthr;
  // the profiling / debug hooks only exist in the HOOKS instantiation: in the product kernel they
  // fold away instead of costing a dozen live scalar registers
  const int ablate = HOOKS ? A.ablate : 0;
  const int Lw = L.w, Lxend = L.xend, Lh = L.h;
  const bool wmod = (Lw & 15) != 0;

  // score rows r = 0 .. R+2  <->  level rows ys-1+r ; image tile row of level row y is y-(ys-4)
  // every pixel that ends up with a non-zero score is also queued as an NMS candidate, so phase D
  // visits ~100 blocks per strip instead of scanning the whole score tile
  auto push_nonzero = [&](bool nz, uint32_t e) {
    const uint64_t m = __ballot(nz);
    if (m) {
      const int cnt = __popcll(m);
      int base = 0;
      if (lane == 0) base = (int)lds_add_rtn(&sh_ctr[2], (uint32_t)cnt);
      base = __builtin_amdgcn_readfirstlane(base);
      if (base + cnt <= QN_SHARED) {
        if (nz) shq_n[base + ballot_rank(m)] = e;
      } else if (lane == 0) {
        sh_ctr[3] = 1;
        sh_ctr[7] |= 2;                             // (reason, for the profiling line)
      }
    }
  };
  // plain layout: corners -> shq_h, Harris writes the score tile and queues the non-zero scores in shq_n.
  // ALIAS layout: ONE queue (shq_h, QH_SHARED entries): FAST appends corners as tile offsets, Harris rewrites the
  // entries appended since `h_begin` in place as x | row << 16 | score << 24 (0xff for over-classified columns), the
  // strip above leaves its carried entries (already in that form) at the front.
  auto harris_batch = [&](bool valid, uint32_t e, int at) {
    uint8_t score = 0;
    if (ALIAS) {
      // the entry is still the corner's tile offset k (see fast_batch): (x, r) by one multiply-high here — a third as
      // many batches as FAST runs — and the window's top-left corner is tile0 + k - 3
      if (valid) {
        const int r = (int)__umulhi(e, L.tp_recip);
        const int x = (int)e - r * tpitch + xbase;
        // Fast.h:172: only x < w-B is scored; over-classified columns keep 0xff
        score = x >= L.xscore ? (uint8_t)0xff : (ablate & 32) ? (uint8_t)200 : harris_score_mm(tile0 + e - 3, tpitch, hthr);
        shq_h[at] = (uint32_t)x | ((uint32_t)r << 16) | ((uint32_t)score << 24);
      }
    } else {
      if (valid) {
        const int x = e & 0xffff, r = (e >> 16) & 0xff;
        score = (ablate & 32) ? (uint8_t)200 : harris_score_mm(tile + r * tpitch + x - 3, tpitch, hthr);
        sc[r * pitch + x] = score;
      }
      push_nonzero(score != 0, e);
    }
  };
  // Candidates travel through the per-wave queues as TILE OFFSETS k = r * tpitch + (x - xbase) (the pixel's byte
  // offset from the first classified row's tile row): the prefilter scans the tile linearly, the pretest and FAST
  // address the pixel as tile3 + k, and (x, r) is recovered by one multiply-high where a corner is queued.
  const lds_u8 *tile3 = tile0 + 3 * tpitch;
  const uint32_t tp_recip = L.tp_recip;
  auto fast_batch = [&](bool valid, uint32_t k) {
    bool corner = false;
    // (a one-sided test keyed on which compass side fired was measured: 16 % of the candidates fire on
    //  both sides, so nearly every 64-lane batch needed the second pass and it was slower)
    if (valid) corner = fast9_mm(tile3 + k, tpitch, thr);
    // ALIAS: the corner is queued as its tile offset; the Harris phase turns it into (x, r, score).  Plain layout:
    // (x, r) is needed here (score tile, NMS queue, over-classified columns: Fast.h:172).
    uint32_t e = k;
    bool toh = corner;
    if (!ALIAS) {
      const int r = (int)__umulhi(k, tp_recip);
      const int x = (int)k - r * tpitch + xbase;
      e = (uint32_t)x | ((uint32_t)r << 16);
      toh = corner && x < L.xscore;
      if (corner && !toh) sc[r * pitch + x] = 0xff;
      push_nonzero(corner && !toh, e);
    }
    const bool q = ALIAS ? corner : toh;
    const uint64_t m = __ballot(q);
    if (m) {
      const int cnt = __popcll(m);
      int base = 0;
      if (lane == 0) base = (int)lds_add_rtn(&sh_ctr[0], (uint32_t)cnt);
      base = __builtin_amdgcn_readfirstlane(base);
      if (base + cnt <= qcap) {
        if (q) shq_h[base + ballot_rank(m)] = e;
      } else {                                     // queue full
        if (lane == 0) {
          atomicMin(&sh_ctr[1], (uint32_t)base);   // entries below `base` stay valid
          sh_ctr[3] = 1;                           // plain: their scores are not queued for NMS -> scan; ALIAS: defer
          sh_ctr[7] |= 1;
        }
        if (!ALIAS) harris_overflow(tile, sc, tpitch, pitch, hthr, toh, e);
      }
    }
  };

  const int r_lo = carry ? 3 : (ys - 1 < B) ? 1 : 0;              // rows above B are never classified
  const int r_hi = min(ye + 2, Lh - B) - (ys - 1);               // exclusive
  // pretest thresholds on h = 128 + floor((N - C) / 2), complemented and replicated (see pretest_batch)
  const uint32_t nkb4 = ~((uint32_t)min(128 + ((thr + 1) >> 1), 255) * 0x01010101u);   // bright: h >= 128 + floor((t + 1) / 2)
  const uint32_t nkd4 = ~((uint32_t)(129 - ((thr + 2) >> 1)) * 0x01010101u);            // not dark: h >= 129 - ceil((t + 1) / 2)
  const bool aligned4 = ((B | Lxend) & 3) == 0;    // the classified range's edges fall on dword boundaries

  // exact compass pretest of the 4 pixels of one group (x0 % 4 == 0), survivors -> qf
  // (the linear prefilter also passes groups of the halo columns: they are dropped here, where x is known)
  const uint32_t cspan = (uint32_t)((cxb + 3) & ~3) - (uint32_t)(cxa & ~3);   // dword groups that hold classified columns
  const int xrel = xbase - (cxa & ~3);
  // (a group travels through the prefilter's queue as the LDS ADDRESS of the dword left of it — the pointer the
  //  prefilter walks with anyway, so that its loop carries no key of its own; the tile offset comes back by one subtract)
  const uint32_t grp_base = (uint32_t)(uintptr_t)(tile3 - 4);
  auto pretest_batch = [&](const bool lane_valid, const uint32_t grp) {
    const uint32_t key = grp - grp_base;
    const int x0r = (int)key - (int)__umulhi(key, tp_recip) * tpitch + xrel;   // x0 - (cxa & ~3)
    const bool valid = lane_valid && (uint32_t)x0r < cspan;
    const int x0 = x0r + (cxa & ~3);
    const lds_u8 *pb = (const lds_u8 *)(uintptr_t)grp;       // (left neighbour first: DS offsets are unsigned)
    const uint32_t wl = *(const lds_u32 *)pb;
    const uint32_t wc = *(const lds_u32 *)(pb + 4);
    const uint32_t wr = *(const lds_u32 *)(pb + 8);
    const uint32_t wu = *(const lds_u32 *)(pb + 4 - 3 * tpitch);
    const uint32_t wd = *(const lds_u32 *)(pb + 4 + 3 * tpitch);
    // The compass test in the BYTE domain, 4 pixels per instruction.  h = v_lerp_u8(N, ~C, 1) = 128 + floor((N - C) / 2)
    // per byte for the neighbour dwords N = up / down / 3 left / 3 right of the group C.  A compass point brighter than
    // C + t has floor((N - C) / 2) >= floor((t + 1) / 2), a darker one <= -ceil((t + 1) / 2): bit 7 of
    // v_lerp_u8(h, ~K, 1) is [h >= K], so eight more lerps against two replicated constants give the four "bright" and
    // the four "not dark" flags, five boolean instructions the pass flag (one vertical AND one horizontal point on the
    // same side).  Halving the difference makes ONE of the two sides one grey level more permissive than the exact
    // test (the bright side for even t, the dark side for odd t): still a necessary condition of the segment test —
    // FAST decides — and 20 instructions where the 16-bit form (two pixels per instruction: ten v_perm_b32 to unpack,
    // min / max / compare on halves) took 32.
    const uint32_t nC = ~wc;
    const uint32_t hU = __builtin_amdgcn_lerp(wu, nC, 0x01010101u), hD = __builtin_amdgcn_lerp(wd, nC, 0x01010101u);
    const uint32_t hL = __builtin_amdgcn_lerp(__builtin_amdgcn_alignbyte(wc, wl, 1), nC, 0x01010101u);   // x-3: l.b1 .. c.b0
    const uint32_t hR = __builtin_amdgcn_lerp(__builtin_amdgcn_alignbyte(wr, wc, 3), nC, 0x01010101u);   // x+3: c.b3 .. r.b2
    const uint32_t bU = __builtin_amdgcn_lerp(hU, nkb4, 0x01010101u), bD = __builtin_amdgcn_lerp(hD, nkb4, 0x01010101u);
    const uint32_t bL = __builtin_amdgcn_lerp(hL, nkb4, 0x01010101u), bR = __builtin_amdgcn_lerp(hR, nkb4, 0x01010101u);
    const uint32_t dU = __builtin_amdgcn_lerp(hU, nkd4, 0x01010101u), dD = __builtin_amdgcn_lerp(hD, nkd4, 0x01010101u);
    const uint32_t dL = __builtin_amdgcn_lerp(hL, nkd4, 0x01010101u), dR = __builtin_amdgcn_lerp(hR, nkd4, 0x01010101u);
    uint32_t bright = (bU | bD) & (bL | bR);
    uint32_t notdark = (dU & dD) | (dL & dR);              // bit 7: no dark vertical point OR no dark horizontal point
#if PISLAM_DIAG
    // A 9-arc also holds two ADJACENT DIAGONAL ring points (positions 3 / 7 / 11 / 15 = (-2,+2), (+2,+2), (+2,-2), (-2,-2):
    // any nine consecutive ring positions contain two consecutive odd multiples-of-2 of either kind), of the arc's own
    // polarity: one of the opposite pair (3, 11) AND one of (7, 15).  Same byte-domain form as the compass points — rows
    // -2 / +2 as three aligned dwords each, the four neighbours of the 4 pixels by v_alignbyte — and the same polarity
    // as the compass pair: 5.3 -> 3.1 candidates per corner on the synthetic input, 4.5 -> 2.4 on the demo photo.
    {
      const lds_u8 *pu2 = pb - 2 * tpitch, *pd2 = pb + 2 * tpitch;
      const uint32_t ul = *(const lds_u32 *)pu2, uc = *(const lds_u32 *)(pu2 + 4), ur = *(const lds_u32 *)(pu2 + 8);
      const uint32_t dl = *(const lds_u32 *)pd2, dc = *(const lds_u32 *)(pd2 + 4), dr = *(const lds_u32 *)(pd2 + 8);
      const uint32_t h15 = __builtin_amdgcn_lerp(__builtin_amdgcn_alignbyte(uc, ul, 2), nC, 0x01010101u);   // (-2, -2)
      const uint32_t h3 = __builtin_amdgcn_lerp(__builtin_amdgcn_alignbyte(ur, uc, 2), nC, 0x01010101u);    // (-2, +2)
      const uint32_t h11 = __builtin_amdgcn_lerp(__builtin_amdgcn_alignbyte(dc, dl, 2), nC, 0x01010101u);   // (+2, -2)
      const uint32_t h7 = __builtin_amdgcn_lerp(__builtin_amdgcn_alignbyte(dr, dc, 2), nC, 0x01010101u);    // (+2, +2)
      const uint32_t b3 = __builtin_amdgcn_lerp(h3, nkb4, 0x01010101u), b11 = __builtin_amdgcn_lerp(h11, nkb4, 0x01010101u);
      const uint32_t b7 = __builtin_amdgcn_lerp(h7, nkb4, 0x01010101u), b15 = __builtin_amdgcn_lerp(h15, nkb4, 0x01010101u);
      const uint32_t d3 = __builtin_amdgcn_lerp(h3, nkd4, 0x01010101u), d11 = __builtin_amdgcn_lerp(h11, nkd4, 0x01010101u);
      const uint32_t d7 = __builtin_amdgcn_lerp(h7, nkd4, 0x01010101u), d15 = __builtin_amdgcn_lerp(h15, nkd4, 0x01010101u);
      bright &= (b3 | b11) & (b7 | b15);
      notdark |= (d3 & d11) | (d7 & d15);
    }
#endif
    const uint32_t pass = bright | ~notdark;               // bit 7 of byte k: pixel x0 + k goes on to the segment test
    // The four pass flags as wave masks, ONE compare each (SDWA: the sign of byte k; groups outside the classified
    // columns cleared first), the early exit decided on the masks, and the lane predicates taken back from them
    // (inverse ballot: no instruction).  (Clearing the masks with scalar ANDs instead of the one v_cndmask was measured:
    // the same VALU count to within 0.1 %, 0.6 M more SALU instructions per launch, not faster.)
    uint64_t m0, m1, m2, m3;
    asm volatile("v_cmp_gt_i32_sdwa %0, 0, sext(%4) src0_sel:DWORD src1_sel:BYTE_0\n\t"
                 "v_cmp_gt_i32_sdwa %1, 0, sext(%4) src0_sel:DWORD src1_sel:BYTE_1\n\t"
                 "v_cmp_gt_i32_sdwa %2, 0, sext(%4) src0_sel:DWORD src1_sel:BYTE_2\n\t"
                 "v_cmp_gt_i32_e64 %3, 0, %4\n\ts_nop 1"
        : "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
        : "v"(valid ? pass : 0u));
    if (!aligned4) {                     // generic border: drop the pixels outside [cxa, cxb)
      m0 &= __ballot(x0 + 0 >= cxa && x0 + 0 < cxb);
      m1 &= __ballot(x0 + 1 >= cxa && x0 + 1 < cxb);
      m2 &= __ballot(x0 + 2 >= cxa && x0 + 2 < cxb);
      m3 &= __ballot(x0 + 3 >= cxa && x0 + 3 < cxb);
    }
    if ((m0 | m1 | m2 | m3) == 0) return;
    const bool p0 = __builtin_amdgcn_inverse_ballot_w64(m0), p1 = __builtin_amdgcn_inverse_ballot_w64(m1);
    const bool p2 = __builtin_amdgcn_inverse_ballot_w64(m2), p3 = __builtin_amdgcn_inverse_ballot_w64(m3);
    // two half-pushes (<= 128 each) with a pop in between keep the queue below 64 + 128 entries
    {
      lds_u32 *q = qf + nf;
      if (p0) q[ballot_rank(m0)] = key;
      q += __popcll(m0);
      if (p1) q[ballot_rank(m1)] = key + 1;
      nf += __popcll(m0) + __popcll(m1);
    }
    while (nf >= 64) {
      nf -= 64;
      if (!(ablate & 2)) fast_batch(true, qf[nf + lane]);
    }
    {
      lds_u32 *q = qf + nf;
      if (p2) q[ballot_rank(m2)] = key + 2;
      q += __popcll(m2);
      if (p3) q[ballot_rank(m3)] = key + 3;
      nf += __popcll(m2) + __popcll(m3);
    }
    while (nf >= 64) {
      nf -= 64;
      if (!(ablate & 2)) fast_batch(true, qf[nf + lane]);
    }
  };

  // ALIAS: entries [0, h_begin) of the queue are already scored (carried from the strip above)
  int h_begin = 0;
  {
    // ---- stage image rows [ys-4, min(ye+6, h)), columns [xbase, xbase + tpitch) -----------------
    {
      const int y_lo = ys - 4;
      const int nrows = min(ye + 6, Lh) - y_lo;
      const int r_st0 = carry_img ? 10 : 0;         // the first 10 rows were carried over
      if (carry_img && (ablate & 2048)) {           // profiling only: no global loads on carried strips
      } else if (VEC16 && carry_img && pf_have) {   // the new rows were prefetched while the strip above ran
        // The tile's rows are contiguous in LDS (tpitch = 16 * vectors per row), so the strip's new rows
        // 10 .. nrows-1 are ONE run of (nrows - 10) * vpr 16-byte vectors: thread t owns vectors t, t + NT, ...
        // — one base address with immediate offsets, one compare per vector, nothing to carry across strips.
        const int nvec = (nrows - 10) * (tpitch >> 4);
        lds_u4 *dstv = (lds_u4 *)(tile0 + 10 * tpitch) + tid;
#pragma unroll
        for (int k = 0; k < PF_MAX; k++)
          if (tid + NT * k < nvec) dstv[NT * k] = pf[k];
      } else if (VEC16) {
        const int vpr = tpitch >> 4;                // 16-byte vectors per row
        // linear vector index i over the rows to stage: LDS offset 16 i (contiguous rows), global offset
        // 16 i + row * (vstep - tpitch) with row = i / vpr by one multiply-high (i < 2^16)
        const int nvec = (nrows - r_st0) * vpr;
        const int gap = A.vstep - tpitch;
        // (32-bit byte offsets: a pyramid is < 2 GiB — checked by the plan — and y_lo, xbase >= 0)
        const uint8_t *src0 = im + (uint32_t)((y_lo + r_st0) * A.vstep + xbase);
        lds_u4 *dstv = (lds_u4 *)(tile0 + r_st0 * tpitch);
        const bool tail = (uint32_t)((y_lo + nrows - 1) * A.vstep + xbase + tpitch) > (uint32_t)lim;
        if (!tail) {
          // All of a thread's loads are issued before the first LDS store (ST_MAX in flight per pass; a strip's
          // R + 10 rows are ~5.3 vectors per thread: one pass).  Written as "load; store" per iteration the compiler
          // waits for every load before its store — 5 memory round trips in a row on the FIRST strip of each run,
          // whose rows no strip above could prefetch: 7.8 k of such a strip's cycles, on the workgroup's critical path.
          constexpr int ST_MAX = 6;
          for (int i0 = tid; i0 < nvec; i0 += NT * ST_MAX) {
            u32x4 st[ST_MAX];
#pragma unroll
            for (int k = 0; k < ST_MAX; k++) {
              const int i = i0 + NT * k;
              if (i < nvec) {
                const int row = (int)__umulhi((uint32_t)i, L.vpr_recip);
                st[k] = *(const u32x4 *)(src0 + (uint32_t)(16 * i + row * gap));   // (a pyramid is < 2 GiB)
              }
            }
#pragma unroll
            for (int k = 0; k < ST_MAX; k++)
              if (i0 + NT * k < nvec) dstv[i0 + NT * k] = st[k];
          }
        } else {
          for (int i = tid; i < nvec; i += NT) {
            // never read past the pyramid buffer (the last tile can overhang the image row: flat
            // addressing like the reference, clipped at the end of the buffer)
            const int row = (int)__umulhi((uint32_t)i, L.vpr_recip);
            const ptrdiff_t off = (ptrdiff_t)(y_lo + r_st0) * A.vstep + xbase + (ptrdiff_t)16 * i + (ptrdiff_t)row * gap;
            u32x4 d;
            if (off + 16 <= lim) {
              d = *(const u32x4 *)(im + off);
            } else {
              uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;   // byte-wise, zero beyond the end (scalars: an indexed array lands in scratch)
#pragma unroll
              for (int k = 0; k < 16; k++) {
                const uint32_t b = off + k < lim ? (uint32_t)im[off + k] << (8 * (k & 3)) : 0u;
                if (k < 4) w0 |= b; else if (k < 8) w1 |= b; else if (k < 12) w2 |= b; else w3 |= b;
              }
              d = (u32x4){w0, w1, w2, w3};
            }
            dstv[i] = d;
          }
        }
      } else {
        for (int i = r_st0 * tpitch + tid; i < nrows * tpitch; i += NT) {
          const int r = i / tpitch, cx = i - r * tpitch;
          const ptrdiff_t off = (ptrdiff_t)(y_lo + r) * A.vstep + xbase + cx;
          tile0[r * tpitch + cx] = off < lim ? im[off] : (uint8_t)0;
        }
      }
    }
    lds_barrier();
    mark(0);
    if (ALIAS) h_begin = (int)sh_ctr[2];            // carried entries (sh_ctr[0] is already being appended to)
    // Prefetch the next strip's R new image rows (level rows ye+6 ..) into registers now: the loads are
    // in flight during this strip's classification and are only waited for when the next strip stores
    // them (lds_barrier does not wait for global loads).
    pf_issued = false;
    if (VEC16 && pf_want && L.R >= 10 && L.R * (tpitch >> 4) <= PF_MAX * NT) {
      const int vpr = tpitch >> 4;
      const int ylo_n = ye - 4, ye_n = min(ye + L.R, Lh - B);
      const int nrows_n = min(ye_n + 6, Lh) - ylo_n;
      if ((uint32_t)((ylo_n + nrows_n - 1) * A.vstep + xbase + tpitch) <= (uint32_t)lim) {
        // the next strip's new rows 10 .. nrows_n-1 as one run of vectors (see the store above)
        const int nvec_n = (nrows_n - 10) * vpr, gap = A.vstep - tpitch;
        const uint8_t *src_n = im + (uint32_t)((ylo_n + 10) * A.vstep + xbase);
#pragma unroll
        for (int k = 0; k < PF_MAX; k++) {
          const int i = tid + NT * k;
          if (i < nvec_n) pf[k] = *(const u32x4 *)(src_n + (uint32_t)(16 * i + (int)__umulhi((uint32_t)i, L.vpr_recip) * gap));
        }
        pf_issued = true;
      }
    }
    if (ablate & 1) return;

    // Group prefilter on every 4-pixel group: a pixel can only pass the compass test if one of its
    // vertical compass points AND one of its horizontal ones differ from it by more than t, so the
    // byte-wise SADs of the group against the rows 3 above/below and the columns 3 left/right must
    // exceed t on both axes (v_sad_u8 sums |a-b| over the 4 bytes, an upper bound of each term).
    // The tile's rows are contiguous in LDS, so the classified rows [r_lo, r_hi) are ONE run of bytes: a wave step
    // covers 256 consecutive bytes of it (one lane = one aligned 4-pixel group, halo columns included — 4 .. 16 %
    // of a row, dropped by the pretest), the steps go round-robin to the waves, and only the run's last step is
    // partial.  Nothing is set up per row, the three pointers and the key advance by constants.
    const int lin_lo = r_lo * tpitch, lin_n = max(r_hi - r_lo, 0) * tpitch;
    const int nfull = lin_n >> 8, rem = lin_n & 255;
    const uint32_t thr1x2 = (uint32_t)(thr + 1) * 0x10001u;
    auto prefilter_step = [&](const lds_u8 *pm, const lds_u8 *pu, const lds_u8 *pd, bool lane_ok) {
      // aligned dword reads; lanes past the tile's columns read harmless bytes of the next tile row
      const uint32_t wl = *(const lds_u32 *)pm;
      const uint32_t wc = *(const lds_u32 *)(pm + 4);
      const uint32_t wr = *(const lds_u32 *)(pm + 8);
      const uint32_t wu = *(const lds_u32 *)pu;
      const uint32_t wd = *(const lds_u32 *)pd;
      // both axes in one register: v_sad_hi_u8 puts its sum into the high half ({left : up}, {right : down}), one packed
      // max gives {max(left, right) : max(up, down)}, an SDWA min of the two halves the value to compare (7 VALU
      // instead of 8: two v_max + v_min)
      const uint32_t a = __builtin_amdgcn_sad_hi_u8(__builtin_amdgcn_alignbyte(wc, wl, 1), wc, __builtin_amdgcn_sad_u8(wu, wc, 0u));
      const uint32_t b = __builtin_amdgcn_sad_hi_u8(__builtin_amdgcn_alignbyte(wr, wc, 3), wc, __builtin_amdgcn_sad_u8(wd, wc, 0u));
      typedef unsigned short us2v __attribute__((ext_vector_type(2)));
      const uint32_t mm = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(us2v, a), __builtin_bit_cast(us2v, b)));
      // both halves > t  <=>  both halves >= t + 1  <=>  v_pk_min_u16(mm, {t+1, t+1}) == {t+1, t+1}: two instructions the
      // compiler schedules itself (the SDWA min of the two halves was an opaque asm statement: an s_nop on either side)
      const uint32_t mn = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us2v, mm), __builtin_bit_cast(us2v, thr1x2)));
      const bool g = lane_ok && (mn == thr1x2);
      const uint64_t m = __ballot(g);
      if (m == 0) return;
      if (g) qg[ng + ballot_rank(m)] = (uint32_t)(uintptr_t)pm;          // the group's address (see pretest_batch)
      ng += __popcll(m);
      if (ng >= 64) {
        ng -= 64;
        if (!(ablate & 16)) pretest_batch(true, qg[ng + lane]);
      }
    };
    // (Unrolling the full steps so that their strides become immediate DS offsets instead of three
    //  pointer increments was tried: every copy of the step inlines the pretest / FAST batch code behind it,
    //  4859 -> 6481 instructions for 3 VALU per step.  Round 4, measured and removed: TWO groups per lane and step
    //  (8-byte centre / up / down reads, 21 VALU + 5 LDS reads per 512 pixels instead of 24 + 10: -1.7 M VALU per
    //  launch, the kernel 1.5 % SLOWER — half as many steps to deal round-robin to four waves); reading the next
    //  step's five dwords before evaluating the current one (the register rotation and re-formed addresses cost
    //  +5.4 M VALU per launch, the kernel +1 us: the latency it hides is worth less than the moves).)
    {
      const lds_u8 *pm = tile3 + lin_lo + 256 * wave + 4 * lane - 4;
      const lds_u8 *pu = pm + 4 - 3 * tpitch, *pd = pm + 4 + 3 * tpitch;
      for (int st = wave; st < nfull; st += WAVES) {
        prefilter_step(pm, pu, pd, true);
        pm += 256 * WAVES;
        pu += 256 * WAVES;
        pd += 256 * WAVES;
      }
      // (after the loop the pointers stand at the first step >= nfull of this wave: the partial step for one wave)
      if (rem && (nfull & (WAVES - 1)) == wave) prefilter_step(pm, pu, pd, 4 * lane < rem);
    }
    if (ng > 0 && !(ablate & 16)) pretest_batch(lane < ng, lane < ng ? qg[lane] : grp_base + (uint32_t)lin_lo);
    ng = 0;
    // left-over candidates (< 64): one partial batch per wave.  (Round 6 measured the alternative — the four waves' left-overs
    // as one list behind a barrier, wave b running batch b of it: 0.12 M of 62.75 M VALU saved per launch, the kernel 1.6 %
    // slower: docs/experiments.md R6.)
    if (nf > 0 && !(ablate & 2)) fast_batch(lane < nf, qf[min(lane, nf - 1)]);
    nf = 0;
    lds_barrier();
    mark(1);
    {
      const int th = (int)min(sh_ctr[0], sh_ctr[1]);
      if (!(ablate & 4))
        for (int c0 = h_begin + wave * 64; c0 < th; c0 += WAVES * 64) {
          const int at = min(c0 + lane, th - 1);
          harris_batch(c0 + lane < th, shq_h[at], at);
        }
    }
    lds_barrier();
    mark(2);
  }
  tile = tile0;
  if ((ablate & 512) && tid == 0) sh_ctr[3] = 1;    // test hook: force the overflow paths
  if (ablate & 512) lds_barrier();
  // ALIAS kernels have no dense fallbacks: a strip whose queues overflowed (very dense corners) is put on
  // the overflow list and redone by k_fused_overflow (non-aliased layout, scan fallbacks) afterwards.
  auto defer = [&]() {
    deferred = true;                                // (every thread: the caller decides on the next strip's carry)
    if (tid == 0) {
      const uint32_t at = atomicAdd(&ovf[0], 1u);
      if (ovf_id != 0xffffffffu) ovf[2 + at] = ovf_id;   // [0] count, [1] count of the previous step, [2..] entries
                                                    // (id 0xffffffff: counted only — k_frame redoes the strip itself)
      if (HOOKS && prof) prof[5] += (sh_ctr[7] & 1 ? 1ull : 0ull) + (sh_ctr[7] & 2 ? 1ull << 20 : 0ull) +
                                    (sh_ctr[7] & 4 ? 1ull << 40 : 0ull);
      sh_ctr[6] = 1;                                // the next strip of the run must not carry from this one
    }
  };
  if (ALIAS) {
    if (sh_ctr[3] != 0) {
      defer();
      return;
    }
    // The image rows 0..R-1 and the per-wave queues are dead now (rows R..R+9 are kept for the next
    // strip of the run): the score tile is laid over them, zeroed, and the scores are scattered into it
    // from the queue entries.
    const int nz = ((L.R + 3) * pitch) >> 4;
    for (int i = tid; i < nz; i += NT) ((lds_u4 *)sc)[i] = (u32x4)(0u);
    lds_barrier();
    const int tn = (int)sh_ctr[0];
    for (int i = tid; i < tn; i += NT) {
      const uint32_t e = shq_h[i];
      if (e >> 24) sc[(int)((e >> 16) & 0xff) * pitch + (int)(e & 0xffff)] = (uint8_t)(e >> 24);
    }
    lds_barrier();
  }

  if (HOOKS && A.dump_score) {   // debug / parity hook: rows this strip owns, [ys, ye)
    uint8_t *dst = score_dump + (size_t)pyr * score_stride + (size_t)L.row0 * A.vstep + L.col0;
    for (int i = tid; i < (ye - ys) * pitch; i += NT) {
      const int r = i / pitch, x = i - r * pitch;
      if (x >= B && x < max(Lxend, wmod ? Lw + 2 : 0)) dst[(ptrdiff_t)(ys + r) * A.vstep + x] = sc[(r + 1) * pitch + x];
    }
  }

  if (ablate & 8) return;
  // ---- phase D (queue-driven): 2x2-block NMS only where a non-zero score exists ---------------
  // Each queued pixel evaluates the block it lies in (Fast.h:228-312) and emits it iff it is the
  // block's winner, so several corners in one block yield exactly one keypoint.  Survivors are
  // ranked by their block-raster key (count of smaller keys) = the reference's push_back order.
  if (sh_ctr[3] == 0) {
    lds_u32 *shq_s = queues;                        // survivors (packed keypoints) in the idle per-wave queues
                                                    // (the tiles stay intact for the next strip of the run)
    lds_u32 *shq_k = shq_s + QS_SHARED;            // their block-raster keys
    const int tn = (int)sh_ctr[ALIAS ? 0 : 2];
    const lds_u32 *nq = ALIAS ? shq_h : shq_n;      // ALIAS: the one queue; entries with score 0 are skipped
    const int own_rows = ye - ys;                   // owned score rows r = 1 .. own_rows
    const int ex0q = L.ex0, ex1q = L.ex1;             // block origins this entry owns
    for (int c0 = wave * 64; c0 < tn; c0 += WAVES * 64) {
      const uint32_t e = nq[min(c0 + lane, tn - 1)];
      const bool valid = c0 + lane < tn && (!ALIAS || (e >> 24) != 0);
      const int x = e & 0xffff, r = (e >> 16) & 0xff;
      const int bx = B + ((x - B) & ~1);            // column of the pixel's 2x2 block
      const bool owned = valid && r >= 1 && r <= own_rows && bx >= ex0q && bx < ex1q;
      // The block rule of Fast.h:228-312 seen from ONE pixel: the block's candidate is the LAST maximum of the block
      // in raster order (s0 / s1 / s2 chain, Fast.h:264-298), and its outer comparisons are >= against the neighbours
      // that precede it in raster order and > against those that follow (Fast.h:265-305) — together exactly "this
      // pixel is the strict maximum of its 3x3 neighbourhood under (score, raster position)".  So an entry decides
      // for itself, on three rows instead of the block's four, four comparisons per instruction:
      //   me >= e for the 4 preceding neighbours  <=>  bit 7 of v_lerp_u8(me, ~e, 1) in every byte,
      //   me >  l for the 4 following ones        <=>  bit 7 of v_lerp_u8(l, ~me, 1) in no byte.
      uint32_t res = 0;
      if (owned) {
        const int cb = (x - 1) & ~3;                // aligned dword holding column x-1
        const uint32_t shb = (uint32_t)(x - 1) & 3u;
        uint32_t w[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const lds_u8 *rowp = sc + (r - 1 + k) * pitch + cb;
          w[k] = __builtin_amdgcn_alignbyte(*(const lds_u32 *)(rowp + 4), *(const lds_u32 *)rowp, shb);   // columns x-1 .. x+2
        }
        const uint32_t me = (w[1] >> 8) & 0xffu;
        const uint32_t rep = me * 0x01010101u;
        const uint32_t before = __builtin_amdgcn_perm(w[1], w[0], 0x04020100u);   // (r-1: x-1, x, x+1), (r: x-1)
        const uint32_t after = __builtin_amdgcn_perm(w[2], w[1], 0x06050402u);    // (r: x+1), (r+1: x-1, x, x+1)
        const uint32_t ge = __builtin_amdgcn_lerp(rep, ~before, 0x01010101u);     // bit 7: me >= that neighbour
        const uint32_t le = __builtin_amdgcn_lerp(after, ~rep, 0x01010101u);      // bit 7: that neighbour >= me
        if (((~ge | le) & 0x80808080u) == 0) res = encode_fast(me, (uint32_t)x, (uint32_t)(ys - 1 + r));
      }
      const uint64_t m = __ballot(res != 0);
      if (m) {
        const int cnt = __popcll(m);
        int base = 0;
        if (lane == 0) base = (int)lds_add_rtn(&sh_ctr[4], (uint32_t)cnt);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base + cnt <= QS_SHARED) {
          if (res) {
            const int slot_i = base + ballot_rank(m);
            shq_s[slot_i] = res;
            shq_k[slot_i] = ((uint32_t)(decode_y(res) - B) >> 1) << 12 | ((uint32_t)(decode_x(res) - B) >> 1);
          }
        } else if (lane == 0) {
          sh_ctr[7] |= 4;
          sh_ctr[3] = 2;                            // too many survivors for the ranking buffer
        }
      }
    }
    lds_barrier();
    mark(3);
    if (sh_ctr[3] == 0) {
      if (ablate & 128) return;
      const int ns = (int)sh_ctr[4];
      const size_t strip_slot = (size_t)pyr * A.slots_per_pyr + L.slot0 + (size_t)s * (L.R >> 1) * L.nbx;
      const uint32_t add_xy = ((uint32_t)L.col0 << 12) | (uint32_t)L.row0;      // README.md:78
      // ---- phase E: orbCompute (Orb.h:396-441) for this strip's keypoints, while the image rows they
      // need (y-15 .. y+15: this strip's rows, the strip above and the rows prefetched for the strip below)
      // are still in the L2 that just served them.  Descriptors go to the strip's staging slots in rank
      // order (rank = position among the strip's keypoints); k_gather_orb moves them to their final place.
      // LDS of this phase (all dead by now): [survivors | ranks | keep][8 patches][vrecpe table].
      const bool orb = ORB && A.orb != 0;
      auto describe_strip = [&](const int n, const uint32_t my_rank) {
        lds_u8 *patches = (lds_u8 *)queues + NMS_SCRATCH * 4;
        lds_u8 *rtab = patches + WAVES * 2 * ORB_PATCH_BYTES;
        lds_barrier();                                // every thread has read the keys it ranks against
        if (tid < n) shq_k[tid] = my_rank;            // (QS_SHARED <= NT: one survivor per thread)
        rtab[tid] = ::g_vrecpe_tab.v[tid & 255];
        lds_barrier();
        const OrbLane G = orb_lane(lane, A.vstep);
        lds_u8 *wave_patches = patches + wave * 2 * ORB_PATCH_BYTES;
        uint32_t *dbase = stage_desc + ((size_t)pyr * A.strips_per_pyr + L.strip0 + s) * (size_t)(QS_SHARED * A.words);
        for (int it = wave; 2 * it < n; it += WAVES) {
          const int i0 = 2 * it, i1 = min(i0 + 1, n - 1);
          const uint32_t r0 = shq_k[i0], r1 = i0 + 1 < n ? shq_k[i1] : 0xffffffffu;
          const uint32_t p0 = r0 != 0xffffffffu ? shq_s[i0] + add_xy : 0u;
          const uint32_t p1 = r1 != 0xffffffffu ? shq_s[i1] + add_xy : 0u;
          if ((p0 | p1) == 0) continue;               // (bucket mode: both dropped)
          const OrbWin w = orb_fetch(G, p0, p1, imb, A.vstep, img_bytes32);
          orb_describe(G, w, p0, p1, wave_patches, A.vstep, (const lds_u8 *)rtab, A.words, (uint8_t *)dbase,
                       (G.half ? r1 : r0) * (uint32_t)A.words);
        }
      };
      if (Albs == 0) {
        uint32_t my_rank = 0;
        if (ns <= 64) {
          // (the usual case, ~17 survivors: one wave, the keys in its lanes, v_readlane instead of an LDS read per
          //  comparison — the loop is a chain of LDS latencies otherwise, on the workgroup's critical path: the
          //  other waves wait for wave 0 at the next strip's first barrier)
          if (wave == 0) {
            const uint32_t key = lane < ns ? shq_k[lane] : 0xffffffffu;
            int rank = 0;
            // four independent comparisons per trip (lanes >= ns hold 0xffffffff: never smaller): a single wave issues
            // about one instruction per five cycles, the loop is this phase's critical path
            int r1 = 0, r2 = 0, r3 = 0;
            for (int j = 0; j < ns; j += 4) {
              rank += (uint32_t)__builtin_amdgcn_readlane((int)key, j) < key;
              r1 += (uint32_t)__builtin_amdgcn_readlane((int)key, j + 1) < key;
              r2 += (uint32_t)__builtin_amdgcn_readlane((int)key, j + 2) < key;
              r3 += (uint32_t)__builtin_amdgcn_readlane((int)key, j + 3) < key;
            }
            rank += r1 + r2 + r3;
            if (lane < ns) stage_kp[strip_slot + rank] = shq_s[lane] + add_xy;
            my_rank = (uint32_t)rank;
          }
        } else {
          for (int i = tid; i < ns; i += NT) {
            const uint32_t key = shq_k[i];
            int rank = 0;
            for (int j = 0; j < ns; j++) rank += shq_k[j] < key;
            stage_kp[strip_slot + rank] = shq_s[i] + add_xy;
            my_rank = (uint32_t)rank;
          }
        }
        if (tid == 0)
          strip_count[(size_t)pyr * A.strips_per_pyr + L.strip0 + s] = (uint32_t)ns | (orb ? STRIP_DESCRIBED : 0u);
        mark(4);
        if (orb && ns > 0) describe_strip(ns, my_rank);
        return;
      }
      // Buckets (Fast.h:314-352): a cell = one bucket x one flush interval = 2^lbs x 2^lbs pixels of
      // block origins; it keeps its `limit` largest packed words, ASCENDING; cells are emitted in
      // (cell-row, bucket) order.  Strip heights are multiples of the cell size, so cells never
      // straddle strips.  Pass 1 marks the survivors that make their cell's top-`limit`, pass 2 ranks
      // the kept ones by (cell, value).
      const int lbs = Albs, limit = A.limit;
      if (ns <= 64) {
        // (the usual case: one wave, the survivors in its lanes, v_readlane instead of two LDS reads per comparison —
        //  written as per-thread loops over LDS the two passes below are chains of ~2 ns LDS round trips on the
        //  workgroup's critical path)
        uint32_t my_rank = 0xffffffffu;
        uint32_t kept = 0;
        if (wave == 0) {
          const uint32_t v = lane < ns ? shq_s[lane] : 0u;
          const uint32_t cell = lane < ns ? ((((uint32_t)(decode_y(v) - B) >> lbs) << 12) | ((uint32_t)(decode_x(v) - B) >> lbs)) : 0xffffffffu;
          // ONE pass over the survivors: how many of the same cell are larger (-> kept iff fewer than `limit`), and how
          // many precede this one in (cell, value) order; then the few DROPPED ones (each cell's smallest) that precede
          // it are taken off again — a loop over the set bits of their ballot, usually none to three.
          int greater = 0, g1 = 0, less = 0, l1 = 0;
          for (int j = 0; j < ns; j += 2) {               // (lanes >= ns hold cell 0xffffffff: never equal, never smaller)
            const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)cell, j), v0 = (uint32_t)__builtin_amdgcn_readlane((int)v, j);
            const uint32_t c1 = (uint32_t)__builtin_amdgcn_readlane((int)cell, j + 1), v1 = (uint32_t)__builtin_amdgcn_readlane((int)v, j + 1);
            greater += (c0 == cell) & (v0 > v);
            g1 += (c1 == cell) & (v1 > v);
            less += (c0 < cell) | ((c0 == cell) & (v0 < v));
            l1 += (c1 < cell) | ((c1 == cell) & (v1 < v));
          }
          greater += g1;
          int rank = less + l1;
          const bool keep = lane < ns && greater < limit;
          const uint64_t km = __ballot(keep);
          uint64_t dm = __ballot(lane < ns && !keep);
          while (dm) {
            const int j = __builtin_ctzll(dm);
            dm &= dm - 1;
            const uint32_t cj = (uint32_t)__builtin_amdgcn_readlane((int)cell, j), vj = (uint32_t)__builtin_amdgcn_readlane((int)v, j);
            rank -= (cj < cell) | ((cj == cell) & (vj < v));
          }
          if (keep) {
            stage_kp[strip_slot + rank] = v + add_xy;
            my_rank = (uint32_t)rank;
          }
          kept = (uint32_t)__popcll(km);
        }
        if (tid == 0) strip_count[(size_t)pyr * A.strips_per_pyr + L.strip0 + s] = kept | (orb ? STRIP_DESCRIBED : 0u);
        mark(4);
        if (orb && ns > 0) describe_strip(ns, my_rank);
        return;
      }
      for (int i = tid; i < ns; i += NT) {
        const uint32_t v = shq_s[i];
        const uint32_t cell = (((uint32_t)(decode_y(v) - B) >> lbs) << 12) | ((uint32_t)(decode_x(v) - B) >> lbs);
        shq_k[i] = cell;
      }
      lds_barrier();
      lds_u32 *keep = shq_k + QS_SHARED;
      for (int i = tid; i < ns; i += NT) {
        const uint32_t v = shq_s[i], cell = shq_k[i];
        int greater = 0;
        for (int j = 0; j < ns; j++) greater += (shq_k[j] == cell) & (shq_s[j] > v);
        keep[i] = greater < limit;
      }
      lds_barrier();
      int kept_here = 0;
      uint32_t my_rank = 0xffffffffu;                 // (dropped by its cell's top-`limit`: not described)
      for (int i = tid; i < ns; i += NT) {
        if (!keep[i]) continue;
        const uint32_t v = shq_s[i], cell = shq_k[i];
        int rank = 0;
        for (int j = 0; j < ns; j++)
          rank += keep[j] & ((shq_k[j] < cell) | ((shq_k[j] == cell) & (shq_s[j] < v)));
        stage_kp[strip_slot + rank] = v + add_xy;
        my_rank = (uint32_t)rank;
        kept_here++;
      }
      if (kept_here) atomicAdd(&sh_ctr[5], (uint32_t)kept_here);
      lds_barrier();
      if (tid == 0) strip_count[(size_t)pyr * A.strips_per_pyr + L.strip0 + s] = sh_ctr[5] | (orb ? STRIP_DESCRIBED : 0u);
      if (orb && ns > 0) describe_strip(ns, my_rank);
      return;
    }
  }
  if constexpr (ALIAS) {
    defer();
    return;
  } else {
  // Fallback (a queue overflowed: very dense corners): scan the whole score tile.
  if (tid == 0) sh_ctr[6] = 1;                      // the fallbacks scribble over the whole image tile
  if (Albs != 0) {
    // Bucket mode: one wave per cell, top-`limit` by repeated wave-max over the cell's blocks.
    const int lbs = Albs, limit = A.limit, bs = 1 << lbs, hb = bs >> 1;
    const int cx0 = (L.ex0 - B) >> lbs;                             // first bucket of this entry (x-tiles start on bucket boundaries)
    const int ncx = (L.ex1 - B - 1) / bs + 1 - cx0;                 // Fast.h:201 numBuckets (of the owned column range)
    const int ncy = (ye - ys + bs - 1) / bs;
    const int ncell = ncx * ncy, nblk = hb * hb;
    const int capc = min(limit, nblk);                              // survivors a cell can keep
    lds_u32 *cellres = (lds_u32 *)tile0;                            // ncell x capc  (<= #blocks dwords)
    lds_u32 *cellcnt = cellres + ncell * capc;                      // ncell
    lds_u32 *cand = queues + wave * QCAP;                           // per-wave scratch: nblk <= 256 dwords
    const int xlimc = L.ex1;
    for (int cell = wave; cell < ncell; cell += WAVES) {
      const int cy = cell / ncx, cx = cx0 + (cell - cy * ncx);
      for (int i = lane; i < nblk; i += 64) {
        const int by = i / hb, bxi = i - by * hb;
        const int x = B + cx * bs + 2 * bxi, y = ys + cy * bs + 2 * by;
        uint32_t r0 = 0;
        if (x < xlimc && y < ye) r0 = nms_block(sc + (y - ys + 1) * pitch + x, pitch, x, y);
        cand[i] = r0;
      }
      int nf_c = 0;
      for (int k = 0; k < capc; k++) {
        uint32_t best = 0;
        for (int i = lane; i < nblk; i += 64) best = max(best, cand[i]);
        best = wave_max_u32(best);
        if (best == 0) break;
        for (int i = lane; i < nblk; i += 64)
          if (cand[i] == best) cand[i] = 0;
        if (lane == 0) cellres[cell * capc + k] = best;               // descending; reversed on emit
        nf_c++;
      }
      if (lane == 0) cellcnt[cell] = (uint32_t)nf_c;
    }
    lds_barrier();
    const size_t strip_slot = (size_t)pyr * A.slots_per_pyr + L.slot0 + (size_t)s * (L.R >> 1) * L.nbx;
    const uint32_t add_xy = ((uint32_t)L.col0 << 12) | (uint32_t)L.row0;
    for (int cell = wave; cell < ncell; cell += WAVES) {
      const uint32_t n = cellcnt[cell];
      if (n == 0) continue;
      uint32_t off = 0;
      for (int k = lane; k < cell; k += 64) off += cellcnt[k];
      off = (uint32_t)wave_sum((int)off);
      if ((uint32_t)lane < n) stage_kp[strip_slot + off + lane] = cellres[cell * capc + (n - 1 - lane)] + add_xy;
    }
    if (wave == 0) {
      uint32_t tot = 0;
      for (int k = lane; k < ncell; k += 64) tot += cellcnt[k];
      tot = (uint32_t)wave_sum((int)tot);
      if (lane == 0) strip_count[(size_t)pyr * A.strips_per_pyr + L.strip0 + s] = tot;
    }
    return;
  }
  // ---- phase D: NMS in ONE pass.  The image tile is dead after the Harris phase, so its LDS is
  // reused as per-block-row survivor buffers: a wave appends its row's survivors in raster order,
  // then (after a barrier) the rows are copied out back to back = block-raster order of the strip.
  __shared__ uint32_t rowcnt[64];
  const int nbr = (ye - ys + 1) >> 1;               // block rows in this strip
  const int xlim = L.ex1, xfirst = L.ex0;          // block origins are x = xfirst, xfirst+2, ... < xlim
  const int nbx = L.nbx;
  lds_u32 *rowbuf = (lds_u32 *)tile0;               // nbr x nbx dwords <= R/2 * w/2 * 4 B < the tile
  // One lane looks at 4 score columns x 2 rows = two horizontally adjacent 2x2 blocks with two
  // aligned dword reads; an all-zero pair (the overwhelmingly common case) is done (Fast.h:237).
  auto nms_pair = [&](const lds_u8 *srow, int x0, int y, uint32_t &ra, uint32_t &rb) {
    ra = rb = 0;
    const uint32_t m = *(const lds_u32 *)(srow + x0) | *(const lds_u32 *)(srow + pitch + x0);
    if (m != 0) {
      // rows y-1..y+2, columns x0-1..x0+2 (left block) and x0+1..x0+4 (right block): three ALIGNED
      // dwords per row + v_alignbyte (unaligned ds_read_b32 stalls ~47 cycles each on gfx950)
      const lds_u8 *p0 = srow - pitch + x0;
      uint32_t a[4], b[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t wl = *(const lds_u32 *)(p0 + k * pitch - 4), wc = *(const lds_u32 *)(p0 + k * pitch),
                       wr = *(const lds_u32 *)(p0 + k * pitch + 4);
        a[k] = __builtin_amdgcn_alignbyte(wc, wl, 3);     // columns x0-1 .. x0+2
        b[k] = __builtin_amdgcn_alignbyte(wr, wc, 1);     // columns x0+1 .. x0+4
      }
      const uint32_t a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
      if (x0 < xlim) ra = nms_block_regs(a0, a1, a2, a3, x0, y);
      if (x0 + 2 < xlim) rb = nms_block_regs(b0, b1, b2, b3, x0 + 2, y);
    }
  };
  const bool pairs = (xfirst & 3) == 0;             // block origins dword-aligned in pairs
  for (int br = wave; br < nbr; br += WAVES) {
    const lds_u8 *srow = sc + (2 * br + 1) * pitch;
    lds_u32 *rb_out = rowbuf + br * nbx;
    uint32_t cnt = 0;
    if (pairs) {
      for (int x0 = xfirst + 4 * lane; x0 - 4 * lane < xlim; x0 += 256) {
        uint32_t ra, rb;
        nms_pair(srow, x0, ys + 2 * br, ra, rb);
        const uint64_t ma = __ballot(ra != 0), mb = __ballot(rb != 0);
        if ((ma | mb) == 0) continue;
        // raster order inside the row: lane-major, then the left block before the right one
        const uint32_t pos = cnt + ballot_rank(ma) + ballot_rank(mb);
        if (ra) rb_out[pos] = ra;
        if (rb) rb_out[pos + (ra != 0)] = rb;
        cnt += __popcll(ma) + __popcll(mb);
      }
    } else {
      for (int bx0 = 0; bx0 < nbx; bx0 += 64) {
        const int bx = bx0 + lane;
        uint32_t res = 0;
        if (xfirst + 2 * bx < xlim) res = nms_block(srow + xfirst + 2 * bx, pitch, xfirst + 2 * bx, ys + 2 * br);
        const uint64_t m = __ballot(res != 0);
        if (res) rb_out[cnt + ballot_rank(m)] = res;
        cnt += __popcll(m);
      }
    }
    if (lane == 0) rowcnt[br] = cnt;
  }
  lds_barrier();
  if (ablate & 128) return;                         // profiling only: NMS compute without the copy-out
  const size_t strip_slot = (size_t)pyr * A.slots_per_pyr + L.slot0 + (size_t)s * (L.R >> 1) * nbx;
  const uint32_t add_xy = ((uint32_t)L.col0 << 12) | (uint32_t)L.row0;      // README.md:78
  for (int br = wave; br < nbr; br += WAVES) {
    const uint32_t n = rowcnt[br];
    if (n == 0) continue;
    uint32_t off = 0;
    for (int k = 0; k < br; k++) off += rowcnt[k];
    for (uint32_t k = lane; k < n; k += 64) stage_kp[strip_slot + off + k] = rowbuf[br * nbx + k] + add_xy;
  }
  if (tid == 0) {
    uint32_t tot = 0;
    for (int k = 0; k < nbr; k++) tot += rowcnt[k];
    strip_count[(size_t)pyr * A.strips_per_pyr + L.strip0 + s] = tot;
  }
  }  // !ALIAS
}

// LDS layouts of the strip kernels (dynamic shared memory):
//   plain  : [image tile (R+10) x tpitch][score tile (R+3) x pitch][per-wave queues][shared queues]
//   ALIAS  : [per-wave queues][apad][image tile (R+10) x tpitch][shared queues]
//            with the score tile laid over [NMS scratch .. image row R) once Harris is done (see
//            strip_body): 26 KB instead of 39 KB per workgroup at VGA level 0 -> more resident workgroups.
struct StripLds {
  lds_u8 *tile, *sc;
  lds_u32 *queues, *shq;
};
template <bool ALIAS>
__device__ __forceinline__ StripLds strip_lds(uint8_t *smem, const FusedLevel &L) {
  StripLds m;
  if (ALIAS) {
    m.queues = (lds_u32 *)smem;
    m.sc = (lds_u8 *)smem + NMS_SCRATCH * 4;
    m.tile = (lds_u8 *)smem + WAVES * QCAP * 4 + L.apad;
    m.shq = (lds_u32 *)(m.tile + (L.R + 10) * L.tpitch);
  } else {
    m.tile = (lds_u8 *)smem;                        // image tile rows [ys-4, ys+R+6), one x-tile at a time
    m.sc = m.tile + L.tbytes;                       // score tile rows [ys-1, ys+R+2), full width
    m.queues = (lds_u32 *)(m.sc + (L.R + 3) * L.pitch);
    m.shq = m.queues + WAVES * QCAP;
  }
  return m;
}

// The strip role of a workgroup: run `run_idx` (launch order: FusedParams::order) of pyramid `pyr`.
// A workgroup walks a RUN of consecutive strips of one level, top to bottom.  From the second strip
// on, the 10 halo image rows and the 3 halo score rows it shares with the strip above are carried
// over inside LDS (strip_body `carry`), so a run behaves like one strip of run_len * R rows at the
// LDS footprint of R rows: the halo is staged, classified and scored once per run, not per strip.
template <bool VEC16, bool HOOKS, bool ALIAS, bool ORBK, bool BUCK, bool INLINE_OVF = false>
__device__ __forceinline__ void strips_role(const FusedParams &P, const uint8_t *__restrict__ pyramids, size_t pyr_stride,
                                            uint32_t *__restrict__ stage_kp, uint32_t *__restrict__ strip_count,
                                            uint8_t *__restrict__ score_dump, size_t score_stride,
                                            unsigned long long *__restrict__ prof, uint32_t *__restrict__ ovf,
                                            uint32_t *__restrict__ stage_desc, uint8_t *smem, uint32_t *sh_ctr, const int pyr,
                                            int run) {
  int li = 0;
  if (P.order_n > 0) {
    const uint32_t o = P.order[run];
    li = (int)(o >> 16);
    run = (int)(o & 0xffffu);
  } else {
    while (li + 1 < P.nlevels && run >= P.lv[li + 1].run0) li++;
    run -= P.lv[li].run0;
  }
  const int s0 = run * P.run_len, s1 = min(s0 + P.run_len, P.lv[li].nstrips);
  long long t_wg = 0;
  if (HOOKS && prof) {
    prof += (size_t)blockIdx.x * 8;                 // one private row of counters per workgroup (zeroed by the host)
    t_wg = clock64();
  }
  bool carry = false, pf_have = false;
  unsigned long long redo = 0;                      // (INLINE_OVF) strips of this run to redo with the plain layout
  u32x4 pf[PF_MAX];
#pragma unroll
  for (int k = 0; k < PF_MAX; k++) pf[k] = (u32x4)(0u);
  for (int s = s0; s < s1; s++) {
    // Opaque copies of the level index, the thread id and the scalar arguments: without them the
    // compiler hoists every strip-invariant address and constant out of this loop and keeps them live
    // across the whole strip body (+38 VGPRs, +90 spilled SGPRs measured); recomputing them per
    // strip costs nothing.
    int li_o = li, tid_o = (int)threadIdx.x;
    asm volatile("" : "+s"(li_o));
    asm volatile("" : "+v"(tid_o));
    const FusedLevel L = P.lv[li_o];
    const StripLds m = strip_lds<ALIAS>(smem, L);
    const uint8_t *im = pyramids + (size_t)pyr * pyr_stride + (size_t)L.row0 * P.vstep + L.col0;
    // bytes of this pyramid's buffer that may be read from the level's origin
    const ptrdiff_t lim = (ptrdiff_t)P.rows * P.vstep - ((ptrdiff_t)L.row0 * P.vstep + L.col0);
    StripArgs A{P.border, P.thr, P.ablate, P.dump_score, P.lbs, P.limit, P.vstep, P.slots_per_pyr,
                P.strips_per_pyr, P.hthr, P.words, P.orb_in_strip};
    asm volatile("" : "+s"(A.border), "+s"(A.thr), "+s"(A.lbs), "+s"(A.limit), "+s"(A.vstep), "+s"(A.slots_per_pyr),
                 "+s"(A.strips_per_pyr), "+s"(A.hthr), "+s"(A.words), "+s"(A.orb));
    const int ys = A.border + s * L.R;              // first block-row y of the strip
    const int ye = min(ys + L.R, L.h - A.border);   // one past the last row owned
    // (OVL: the strips of a run follow each other without a barrier — see strip_body; counters double-buffered by parity)
    constexpr bool OVL = PISLAM_OVL && ALIAS && !BUCK && !ORBK;
    uint32_t *ctr = sh_ctr + (OVL ? 8 * (s & 1) : 0), *ctr_prev = sh_ctr + (OVL ? 8 * ((s & 1) ^ 1) : 0);
    bool deferred = false;
    strip_body<VEC16, HOOKS, ALIAS, ORBK, BUCK>(A, L, pyr, s, ys, ye, m.tile, m.sc, m.queues, m.shq, ctr, ctr_prev, deferred, im, lim,
                                    stage_kp, strip_count, score_dump, score_stride, carry, tid_o, prof, pf, pf_have, s + 1 < s1,
                                    pf_have, ovf, INLINE_OVF ? 0xffffffffu : ((uint32_t)pyr << 16) | (uint32_t)(L.strip0 + s), stage_desc,
                                    pyramids + (size_t)pyr * pyr_stride, (uint32_t)((size_t)P.rows * P.vstep));
    // k_frame: a strip whose queues overflowed is redone by this workgroup with the plain layout, AFTER the run (below)
    if (INLINE_OVF && ALIAS && deferred) redo |= 1ull << (s - s0);
    if (s + 1 < s1) {
      if (OVL && !(HOOKS && (P.ablate & 0xfbf))) {   // (ablations that cut phases keep the barriers)
        // a deferred strip left the body early (its waves are not aligned on a barrier) and leaves no scores: the next
        // strip starts afresh, behind a barrier
        if (deferred) lds_barrier();
        carry = !deferred && L.R >= 10;
      } else {
        lds_barrier();                              // every read of this strip's LDS state is done
        // (a scan fallback scribbles over the tiles, a deferred strip leaves no scores: start afresh)
        carry = ctr[6] == 0 && L.R >= 10 && !(HOOKS && (P.ablate & 1024));
        lds_barrier();
      }
      if (HOOKS && prof && threadIdx.x == 0) prof[6] += 1ull;
    }
  }
  if (HOOKS && prof && threadIdx.x == 0) prof[7] += (unsigned long long)(clock64() - t_wg);
  // k_frame: the strips of this run whose queues overflowed (dense noise, checkerboards) are redone here with the plain
  // layout and its scan fallbacks — what k_fused_overflow does in the three-launch path; the launch's dynamic LDS covers
  // both layouts.  A loop of its own BEHIND the run: inlined into the run's loop body (round 5) the second strip body
  // cost k_frame 42 spilled SGPRs and a private segment on its hot path.
  if (INLINE_OVF && ALIAS) {
    redo = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)redo)) |
           ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(redo >> 32)) << 32);
    while (redo) {
      const int s = s0 + __builtin_ctzll(redo);
      redo &= redo - 1;
      lds_barrier();
      int li_o = li, tid_o = (int)threadIdx.x;
      asm volatile("" : "+s"(li_o));
      asm volatile("" : "+v"(tid_o));
      const FusedLevel L = P.lv[li_o];
      const StripLds m2 = strip_lds<false>(smem, L);
      const uint8_t *im = pyramids + (size_t)pyr * pyr_stride + (size_t)L.row0 * P.vstep + L.col0;
      const ptrdiff_t lim = (ptrdiff_t)P.rows * P.vstep - ((ptrdiff_t)L.row0 * P.vstep + L.col0);
      const StripArgs A{P.border, P.thr, 0, 0, P.lbs, P.limit, P.vstep, P.slots_per_pyr, P.strips_per_pyr, P.hthr, P.words, 0};
      const int ys = A.border + s * L.R, ye = min(ys + L.R, L.h - A.border);
      bool d2 = false, issued = false;
      strip_body<VEC16, false, false, false, true>(A, L, pyr, s, ys, ye, m2.tile, m2.sc, m2.queues, m2.shq, sh_ctr, sh_ctr, d2, im, lim,
                                                   stage_kp, strip_count, nullptr, 0, false, tid_o, nullptr, pf, false, false, issued,
                                                   nullptr, 0u, nullptr, nullptr, 0u);
    }
  }
}

template <bool VEC16, bool HOOKS, bool ALIAS, bool ORBK = false, bool BUCK = true>
__global__ __launch_bounds__(NT) void k_fused_strips(
    const FusedParams P, const uint8_t *__restrict__ pyramids, size_t pyr_stride,
    uint32_t *__restrict__ stage_kp, uint32_t *__restrict__ strip_count,
    uint8_t *__restrict__ score_dump, size_t score_stride, unsigned long long *__restrict__ prof,
    uint32_t *__restrict__ ovf, uint32_t *__restrict__ stage_desc) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ uint32_t sh_ctr[16];                   // two blocks of counters: strips of a run alternate (strips_role)
  // XCD-aware mapping: workgroup b runs on XCD b%8; keep all strips of one pyramid on one XCD so
  // the halo rows shared by neighbouring runs are served by that XCD's L2.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  // Run-major order: all pyramids' run 0 first, ... the short runs of the small levels last, so the
  // tail of the grid consists of short workgroups (a run is up to run_len strips long).
  // (A resident grid pulling runs from per-XCD atomic counters was measured: no gain at any batch.)
  const int groups = (P.batch + 7) >> 3;
  const int pyr = (slot % groups) * 8 + xcd;
  if (pyr >= P.batch) return;
  strips_role<VEC16, HOOKS, ALIAS, ORBK, BUCK>(P, pyramids, pyr_stride, stage_kp, strip_count, score_dump, score_stride, prof, ovf,
                                         stage_desc, smem, sh_ctr, pyr, slot / groups);
}

// Strips the ALIAS kernel could not finish (a queue overflowed) are redone here, one strip per
// workgroup iteration, with the plain layout and its scan fallbacks.  Launched after every ALIAS
// launch with a small fixed grid; normally the list is empty and the workgroups exit at once.
template <bool VEC16, bool HOOKS>
__global__ __launch_bounds__(NT) void k_fused_overflow(
    const FusedParams P, const uint8_t *__restrict__ pyramids, size_t pyr_stride,
    uint32_t *__restrict__ stage_kp, uint32_t *__restrict__ strip_count,
    uint8_t *__restrict__ score_dump, size_t score_stride, const uint32_t *__restrict__ ovf) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ uint32_t sh_ctr[8];
  const uint32_t n = ovf[0];
  u32x4 pf[PF_MAX];
#pragma unroll
  for (int k = 0; k < PF_MAX; k++) pf[k] = (u32x4)(0u);
  for (uint32_t it = blockIdx.x; it < n; it += gridDim.x) {
    const uint32_t id = ovf[2 + it];
    int pyr_o = (int)(id >> 16), sg_o = (int)(id & 0xffffu), tid_o = (int)threadIdx.x;
    asm volatile("" : "+s"(pyr_o), "+s"(sg_o));
    asm volatile("" : "+v"(tid_o));
    if (pyr_o >= P.batch || sg_o >= P.strips_per_pyr) continue;   // stale entry of an aborted launch
    int li = 0;
    while (li + 1 < P.nlevels && sg_o >= P.lv[li + 1].strip0) li++;
    const FusedLevel L = P.lv[li];
    const int s = sg_o - L.strip0;
    const StripLds m = strip_lds<false>(smem, L);
    const uint8_t *im = pyramids + (size_t)pyr_o * pyr_stride + (size_t)L.row0 * P.vstep + L.col0;
    const ptrdiff_t lim = (ptrdiff_t)P.rows * P.vstep - ((ptrdiff_t)L.row0 * P.vstep + L.col0);
    // (ablate bit 512 forces the ALIAS kernel to defer; here it would force the scan fallbacks too — keep it)
    const StripArgs A{P.border, P.thr, P.ablate, P.dump_score, P.lbs, P.limit, P.vstep, P.slots_per_pyr,
                      P.strips_per_pyr, P.hthr, P.words, 0};
    const int ys = A.border + s * L.R;
    const int ye = min(ys + L.R, L.h - A.border);
    bool issued = false;
    bool deferred = false;
    strip_body<VEC16, HOOKS, false, false, true>(A, L, pyr_o, s, ys, ye, m.tile, m.sc, m.queues, m.shq, sh_ctr, sh_ctr, deferred, im, lim, stage_kp,
                                    strip_count, score_dump, score_stride, false, tid_o, nullptr, pf, false, false, issued,
                                    nullptr, 0u, nullptr, nullptr, 0u);
    lds_barrier();
  }
}

// ---------------------------------------------------------------------------
// From strip lists to the reference's keypoint order (both gather kernels).
// Plan strips are stored entry-major (plan entry, then strip top to bottom); `soff` is the exclusive prefix of
// their counts in that order.  For a whole-level entry that IS the reference's push_back order (level order,
// block-raster inside a level, Fast.h:228-320).  A level cut into x-tiles (FusedLevel::gn > 1) emits, per
// strip, one list per tile, each in block-raster order of its own columns; the level's order interleaves them
// by block row (bucket mode: by cell row, Fast.h:211-226): tile t's keypoint of row-key r comes after every
// keypoint of tiles < t with key <= r and of tiles > t with key < r.
// ---------------------------------------------------------------------------
struct PlanStrip {
  int li;            // plan entry
  int s;             // strip index inside the entry
};
__device__ __forceinline__ PlanStrip plan_strip(const FusedParams &P, int i) {
  int li = 0;
  while (li + 1 < P.nlevels && i >= P.lv[li + 1].strip0) li++;
  return {li, i - P.lv[li].strip0};
}
__device__ __forceinline__ uint32_t strip_slot_of(const FusedParams &P, int li, int s) {
  return (uint32_t)(P.lv[li].slot0 + s * (P.lv[li].R >> 1) * P.lv[li].nbx);
}
// final index (within the pyramid) of the k-th keypoint `v` of plan strip (li, s); soff in LDS or global,
// stage = this pyramid's staging slots
template <class OFF>
__device__ __forceinline__ uint32_t final_position(const FusedParams &P, const OFF *soff, const uint32_t *__restrict__ stage,
                                                   int li, int s, uint32_t k, uint32_t v) {
  const int gn = P.lv[li].gn;
  if (gn == 1) return soff[P.lv[li].strip0 + s] + k;
  const int g0 = P.lv[li].gfirst;
  const int shift = P.lbs ? P.lbs : 1;
  const int r = (decode_y(v) - P.lv[li].row0 - P.border) >> shift;      // block row (cell row) inside the level
  uint32_t pos = soff[P.lv[g0].strip0] + k;       // everything before this level ...
  for (int t = 0; t < gn; t++) {
    const int e = g0 + t, j = P.lv[e].strip0 + s;
    pos += soff[j] - soff[P.lv[e].strip0];          // ... + the level's strips above this one (all tiles)
    if (e == li) continue;
    // keypoints of tile t's list that precede: row key < r, or == r when the tile lies to the left
    const int lim = r + (e < li ? 1 : 0);
    const uint32_t *list = stage + strip_slot_of(P, e, s);
    uint32_t lo = 0, hi = soff[j + 1] - soff[j];
    while (lo < hi) {
      const uint32_t m = (lo + hi) >> 1;
      if (((decode_y(list[m]) - P.lv[e].row0 - P.border) >> shift) < lim) lo = m + 1; else hi = m;
    }
    pos += lo;
  }
  return pos;
}

// One workgroup per pyramid: exclusive scan of the strip counts, copy staged keypoints to their final
// positions, publish the total (layouts the fused gather + ORB kernel does not take: vstep % 16 != 0).
__global__ __launch_bounds__(256) void k_gather(const FusedParams P,
                                                const uint32_t *__restrict__ stage_kp,
                                                const uint32_t *__restrict__ strip_count,
                                                uint32_t *__restrict__ kp, size_t kp_stride,
                                                uint32_t cap, uint32_t *__restrict__ counts,
                                                uint32_t *__restrict__ ovf_reset) {
  extern __shared__ uint32_t soff[];                // strips_per_pyr + 1
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t carry;
  // the overflow list of this step has been consumed (stream order): empty it for the next step
  if (ovf_reset && blockIdx.x == 0 && threadIdx.x == 0) {
    ovf_reset[1] = ovf_reset[0];                     // kept for pislam_frontend_last_stats
    ovf_reset[0] = 0;
  }
  const int pyr = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int S = P.strips_per_pyr;
  const uint32_t *cnt = strip_count + (size_t)pyr * S;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < S; base += 256) {
    const int i = base + tid;
    const uint32_t v = i < S ? (cnt[i] & ~STRIP_DESCRIBED) : 0;
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
  const uint32_t *stage = stage_kp + (size_t)pyr * P.slots_per_pyr;
  // one wave per strip
  for (int st = wv; st < S; st += 4) {
    const uint32_t n = soff[st + 1] - soff[st];
    if (n == 0) continue;
    const PlanStrip ps = plan_strip(P, st);
    const uint32_t *list = stage + strip_slot_of(P, ps.li, ps.s);
    for (uint32_t k = lane; k < n; k += 64) {
      const uint32_t v = list[k];
      const uint32_t pos = final_position(P, soff, stage, ps.li, ps.s, k, v);
      if (pos < cap) kp[(size_t)pyr * kp_stride + pos] = v;
    }
  }
}

// ===========================================================================
// k_bucket_select — fastExtract's buckets (Fast.h:200-226, 314-352) as a pass of their own between the strip kernel
// and the gather: the strips run exactly as without buckets (plain block-raster lists per strip, any strip height),
// and ONE WAVE per unit = (pyramid, level, cell row = 2^lbs rows of block origins) collects the unit's NMS survivors
// from the lists of the strips that overlap its rows (all x-tiles of the level), keeps each bucket's `limit` largest
// packed words, and writes them in the reference's flush order — bucket index, then ascending word (Fast.h:321-352) —
// to the unit's slots of a second staging area.  k_gather_orb / k_gather then run on a UNIT plan (one "strip" per
// unit, lists already final, no tiles), i.e. concatenate.  A cell row never depends on another: no state is carried.
//   bucket of a survivor = (x - B) >> lbs and cell row = (y - B) >> lbs of the WINNER pixel (level-relative): the
//   reference uses the block origin (Fast.h:316), which differs by at most one in each coordinate and never across a
//   bucket boundary (origins are even offsets from B, 2^lbs is even).  Packed words compare alike in stacked and
//   level-relative coordinates (the same offsets are added to every word of a level).
// ===========================================================================
struct SelectPlan {
  int nlevels, lbs, limit, border;
  int units_per_pyr, uslots_per_pyr;
  int row0[16], col0[16], h[16];      // the level (not its tiles)
  int g0[16], gn[16];                 // its plan entries in the strip plan
  int unit0[16], nunits[16];          // its units (cell rows)
  int cap[16], uslot0[16];            // slots per unit (buckets x limit), first slot of the level
  int nb_max;                         // most buckets of any level (the dense path's LDS counters: 2 x nb_max dwords per wave)
};
// (k_bucket_select takes FusedParams + SelectPlan + four pointers by value: the kernel-argument segment holds 4 KiB)
static_assert(sizeof(FusedParams) + sizeof(SelectPlan) + 5 * sizeof(void *) <= 4096, "k_bucket_select's arguments exceed the 4 KiB kernarg segment");
constexpr int SEL_WAVES = 4;
constexpr int SEL_NB = 1024;                         // buckets per cell row the dense path's LDS counters hold (the host checks)
// The unit table (host-built, one record of SEL_REC dwords per unit = (level, cell row)): everything the fast path needs to
// know about a unit arrives with ONE scalar load — until round 5 a wave found its level by a loop of dependent scalar
// loads over the plan, and the strips overlapping its rows by two software divisions per x-tile with more plan loads
// in between: the kernel is a single round of resident waves, its duration IS that chain.
//   [0] lists overlapping the unit (0xffffffff: more than SEL_ML: the generic path), [1] level, [2] cell row,
//   [3] first output slot (uslot0 + cell row * cap), [4] row0 + B, [5] col0 + B (stacked coordinates of the level's first
//   block origin), [8 ..] staging slot of list j, [16 ..] strip-count index of list j.
constexpr int SEL_REC = 24, SEL_ML = 6;
__global__ __launch_bounds__(64 * SEL_WAVES) void k_bucket_select(const FusedParams F, const SelectPlan Q,
                                                                   const uint32_t *__restrict__ stage_kp,
                                                                   const uint32_t *__restrict__ strip_count,
                                                                   uint32_t *__restrict__ ustage, uint32_t *__restrict__ ucount,
                                                                   const uint32_t *__restrict__ utab) {
  // dynamic LDS, per wave: [64 survivors of the fast path][nb_max per-bucket counts][nb_max kept entries before the bucket]
  // (sized by the host from the level table: static arrays for the worst case cost the kernel its occupancy — it is a
  //  chain of dependent loads per wave, its duration is the number of rounds of resident waves)
  extern __shared__ uint32_t sel_lds[];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint32_t *sbuf = sel_lds + (size_t)wv * (64 + 2 * Q.nb_max);
  uint32_t *bcnt = sbuf + 64, *bpre = bcnt + Q.nb_max;
  const int pyr = blockIdx.y, u = (int)blockIdx.x * SEL_WAVES + wv;
  if (u >= Q.units_per_pyr) return;
  const uint32_t *rec = utab + (size_t)u * SEL_REC;            // (wave-uniform address: scalar loads)
  const int nl_rec = (int)rec[0], l = (int)rec[1], cr = (int)rec[2];
  const int lbs = Q.lbs, limit = Q.limit, B = Q.border;
  const int y0 = B + (cr << lbs), y1 = min(y0 + (1 << lbs), Q.h[l] - B);        // level-relative rows of the unit
  const int yrow0 = (int)rec[4] - B, xcol0 = (int)rec[5];
  const uint32_t *stage = stage_kp + (size_t)pyr * F.slots_per_pyr;
  const uint32_t *cnt = strip_count + (size_t)pyr * F.strips_per_pyr;
  uint32_t *out = ustage + (size_t)pyr * Q.uslots_per_pyr + rec[3];
  // every chunk of 64 staged entries of the strips that overlap the unit's rows: f(v, in) with in = the lane holds a
  // survivor of this unit   (generic path: dense units, more than SEL_ML lists)
  auto for_each_chunk = [&](auto f) {
    for (int t = 0; t < Q.gn[l]; t++) {
      const int e = Q.g0[l] + t, R = F.lv[e].R;
      const int s_lo = (y0 - B) / R, s_hi = min((y1 - 1 - B) / R, F.lv[e].nstrips - 1);
      for (int sidx = s_lo; sidx <= s_hi; sidx++) {
        const uint32_t n = cnt[F.lv[e].strip0 + sidx] & ~STRIP_DESCRIBED;
        const uint32_t *list = stage + F.lv[e].slot0 + (size_t)sidx * (R >> 1) * F.lv[e].nbx;
        for (uint32_t q0 = 0; q0 < n; q0 += 64) {
          const uint32_t q = q0 + (uint32_t)lane;
          const uint32_t v = q < n ? list[q] : 0u;
          // (cell row of the winner pixel = cell row of its block origin: see above)
          f(v, q < n && (int)((uint32_t)(decode_y(v) - yrow0 - B) >> lbs) == cr);
        }
      }
    }
  };
  auto bucket_of = [&](uint32_t v) -> uint32_t { return (uint32_t)(decode_x(v) - xcol0) >> lbs; };
  // ---- collect (at most 64: the fast path keeps them in the wave's lanes) ----
  int n_unit = 0;
  auto collect = [&](uint32_t v, bool in) {
    const uint64_t m = __ballot(in);
    if (in) {
      const int at = n_unit + ballot_rank(m);
      if (at < 64) sbuf[at] = v;
    }
    n_unit += __popcll(m);
  };
  // The lists that overlap the unit (one to three strips of one to three tiles) come from the record; their counts (scalar
  // loads) and the first 64 entries of every list are requested together, before any of them is looked at (a strip's
  // slots exist whatever its count — the staging buffer carries 64 dwords of slack past its last strip): one memory
  // round trip behind the record's.
  constexpr int ML = SEL_ML;
  if (nl_rec >= 0 && nl_rec <= ML) {
    const int nl = nl_rec;
    uint32_t vj[ML], nj[ML];
#pragma unroll
    for (int j = 0; j < ML; j++)
      if (j < nl) {
        nj[j] = cnt[rec[16 + j]] & ~STRIP_DESCRIBED;
        vj[j] = stage[rec[8 + j] + (uint32_t)lane];
      }
#pragma unroll
    for (int j = 0; j < ML; j++)
      if (j < nl) {
        const uint32_t n = nj[j];
        collect(vj[j], (uint32_t)lane < n && (int)((uint32_t)(decode_y(vj[j]) - yrow0 - B) >> lbs) == cr);
        const uint32_t *list = stage + rec[8 + j];
        for (uint32_t q0 = 64; q0 < n; q0 += 64) {                               // (a list of more than 64 entries: dense input)
          const uint32_t q = q0 + (uint32_t)lane;
          const uint32_t v = q < n ? list[q] : 0u;
          collect(v, q < n && (int)((uint32_t)(decode_y(v) - yrow0 - B) >> lbs) == cr);
        }
      }
  } else {
    for_each_chunk(collect);
  }
  if (n_unit == 0) {
    if (lane == 0) ucount[(size_t)pyr * Q.units_per_pyr + u] = 0;
    return;
  }
  if (n_unit <= 64) {
    const uint32_t v = lane < n_unit ? sbuf[lane] : 0u;                    // (same wave wrote them: no barrier)
    const uint32_t b = lane < n_unit ? bucket_of(v) : 0xffffffffu;
    // one pass: larger words of the same bucket (kept iff fewer than `limit`), entries that precede in (bucket, word)
    // order; then the dropped ones (each bucket's smallest) that precede are taken off again
    int greater = 0, g1 = 0, less = 0, l1 = 0;
    for (int j = 0; j < n_unit; j += 2) {                                        // (lanes >= n_unit: bucket 0xffffffff)
      const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)b, j), v0 = (uint32_t)__builtin_amdgcn_readlane((int)v, j);
      const uint32_t b1 = (uint32_t)__builtin_amdgcn_readlane((int)b, j + 1), v1 = (uint32_t)__builtin_amdgcn_readlane((int)v, j + 1);
      greater += (b0 == b) & (v0 > v);
      g1 += (b1 == b) & (v1 > v);
      less += (b0 < b) | ((b0 == b) & (v0 < v));
      l1 += (b1 < b) | ((b1 == b) & (v1 < v));
    }
    greater += g1;
    int rank = less + l1;
    const bool keep = lane < n_unit && greater < limit;
    const uint64_t km = __ballot(keep);
    uint64_t dm = __ballot(lane < n_unit && !keep);
    while (dm) {
      const int j = __builtin_ctzll(dm);
      dm &= dm - 1;
      const uint32_t bj = (uint32_t)__builtin_amdgcn_readlane((int)b, j), vj = (uint32_t)__builtin_amdgcn_readlane((int)v, j);
      rank -= (bj < b) | ((bj == b) & (vj < v));
    }
    if (keep) out[rank] = v;
    if (lane == 0) ucount[(size_t)pyr * Q.units_per_pyr + u] = (uint32_t)__popcll(km);
    return;
  }
  // ---- dense unit (more than 64 survivors): per-bucket counts in LDS, then every survivor against all others ----
  const int ncx = Q.cap[l] / limit;                                             // buckets of this level (<= SEL_NB: the host checks)
  for (int i = lane; i < ncx; i += 64) bcnt[i] = 0;
  for_each_chunk([&](uint32_t v, bool in) {
    if (in) atomicAdd(&bcnt[bucket_of(v)], 1u);
  });
  // kept entries of the buckets before bucket i: exclusive prefix of min(count, limit) — SEL_NB / 64 buckets per lane
  constexpr int PER = SEL_NB / 64;
  uint32_t sum = 0;
  for (int k = 0; k < PER; k++) {
    const int i = PER * lane + k;
    sum += i < ncx ? min(bcnt[i], (uint32_t)limit) : 0u;
  }
  uint32_t incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)incl, d, 64);
    if (lane >= d) incl += t;
  }
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  uint32_t run = incl - sum;
  for (int k = 0; k < PER; k++) {
    const int i = PER * lane + k;
    if (i < ncx) {
      bpre[i] = run;
      run += min(bcnt[i], (uint32_t)limit);
    }
  }
  for_each_chunk([&](uint32_t v, bool in) {
    const uint32_t b = in ? bucket_of(v) : 0xffffffffu;
    int greater = 0, smaller = 0;
    for_each_chunk([&](uint32_t w, bool win) {
      const uint32_t bw = win ? bucket_of(w) : 0xfffffffeu;
      for (int j = 0; j < 64; j++) {
        const uint32_t bj = (uint32_t)__builtin_amdgcn_readlane((int)bw, j), wj = (uint32_t)__builtin_amdgcn_readlane((int)w, j);
        greater += (bj == b) & (wj > v);
        smaller += (bj == b) & (wj < v);
      }
    });
    if (in && greater < limit) {
      const uint32_t c = bcnt[b];
      out[bpre[b] + (uint32_t)smaller - (c - min(c, (uint32_t)limit))] = v;
    }
  });
  if (lane == 0) ucount[(size_t)pyr * Q.units_per_pyr + u] = total;
}

// Small shared state of the gather + ORB role, carved from the FRONT of the dynamic LDS (not static: the strip
// role's 5-workgroups-per-CU budget has no room for another 300 static bytes).
struct OrbShared {
  uint32_t wsum[4];
  uint32_t carry, ntodo, flag, pad;
  uint8_t rtab[256];                                // vrecpe estimate table (256 threads: one entry each)
};
__host__ __device__ constexpr size_t orb_lds_bytes(int strips_per_pyr, size_t per_max) {
  return sizeof(OrbShared) + (size_t)OWAVES * 2 * ORB_PATCH_BYTES + sizeof(uint32_t) * (((size_t)strips_per_pyr + 1 + 3) & ~(size_t)3) +
         sizeof(uint32_t) * (((size_t)strips_per_pyr + 3) & ~(size_t)3) +   // where each strip's list starts in the staging buffer
         sizeof(uint32_t) * 2 * per_max;            // keypoints to describe here and their final positions
}

// The gather + ORB role of a workgroup: chunk `ch` of `nch` of pyramid `pyr` (see k_gather_orb).
struct NoWait {
  __device__ __forceinline__ bool operator()() const { return true; }
};
// `wait_producers` (k_frame): called once the loads that do not depend on the strips (vrecpe table, mask-table row) are
// under way; returns false (workgroup-uniform) when the lists never arrived — the role then does nothing.
template <class WAIT = NoWait, bool GHOOKS = false>
__device__ __forceinline__ void orb_role(const FusedParams &P, const uint8_t *__restrict__ pyramids, size_t pyr_stride,
                                         const uint32_t *__restrict__ stage_kp, const uint32_t *__restrict__ strip_count,
                                         const uint32_t *__restrict__ stage_desc, uint32_t *__restrict__ kp, size_t kp_stride,
                                         uint32_t cap, uint32_t *__restrict__ counts, uint32_t *__restrict__ desc,
                                         size_t desc_stride, int words, uint32_t per_max, uint8_t *osm_all, const int pyr,
                                         const int ch, const int nch, WAIT wait_producers = WAIT()) {
  OrbShared *sh = (OrbShared *)osm_all;
  uint8_t *osm = osm_all + sizeof(OrbShared);
  sh->rtab[threadIdx.x] = ::g_vrecpe_tab.v[threadIdx.x];
  const lds_u8 *rtab = (const lds_u8 *)sh->rtab;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = P.strips_per_pyr;
  // (the lane's mask-table row is loaded FIRST: its memory round trip then runs under the strip-count loads of the
  //  scan instead of standing between the gather part and the first patch fetch)
  const OrbLane G = orb_lane(lane, P.vstep);
  if (!wait_producers()) return;
  // LDS carve: patches (4 waves x 2 x 1.5 KiB) | strip offsets (S+1) | this round's keypoints still to describe |
  // their final positions
  uint8_t *patches = osm;
  uint32_t *soff = (uint32_t *)(osm + OWAVES * 2 * ORB_PATCH_BYTES);
  uint32_t *sslot = soff + ((S + 1 + 3) & ~3);
  uint32_t *kpl = sslot + ((S + 3) & ~3);
  uint32_t *kpos = kpl + per_max;
  // Per strip, once: the staging slot its list starts at (bit 31: its level is cut into x-tiles, the final position needs
  // final_position).  The plan lives in the kernel arguments; indexed PER KEYPOINT it was a divergent loop over the entries
  // plus two rounds of dependent per-lane global loads from the argument segment in front of every keypoint's own load —
  // the longest latency chain of the workgroup's prologue.  Independent of the counts: runs under their loads.
  for (int i = tid; i < S; i += 256) {
    const PlanStrip ps = plan_strip(P, i);
    sslot[i] = strip_slot_of(P, ps.li, ps.s) | (P.lv[ps.li].gn > 1 ? 0x80000000u : 0u);
  }

  // ---- exclusive scan of the strip counts (storage order) ----
  const uint32_t *cnt = strip_count + (size_t)pyr * S;
  if (tid == 0) sh->carry = 0;
  __syncthreads();
  for (int base = 0; base < S; base += 256) {
    const int i = base + tid;
    const uint32_t v = i < S ? (cnt[i] & ~STRIP_DESCRIBED) : 0;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t t = (uint32_t)__shfl_up((int)incl, d, 64);
      if (lane >= d) incl += t;
    }
    if (lane == 63) sh->wsum[wv] = incl;
    __syncthreads();
    uint32_t pre = sh->carry;
    for (int w = 0; w < wv; w++) pre += sh->wsum[w];
    if (i < S) soff[i] = pre + incl - v;
    __syncthreads();
    if (tid == 255) sh->carry = pre + incl;
    __syncthreads();
  }
  const uint32_t total = sh->carry;
  if (tid == 0) {
    soff[S] = total;
    if (ch == 0) counts[pyr] = total;
  }
  __syncthreads();
  const uint32_t per = (total + nch - 1) / nch;
  const uint32_t lo = min((uint32_t)ch * per, total), hi = min(lo + per, total);
  if (lo >= hi) return;

  const uint32_t *stage = stage_kp + (size_t)pyr * P.slots_per_pyr;
  const uint32_t *sd = stage_desc + (size_t)pyr * S * QS_SHARED * words;
  uint32_t *dsc = desc + (size_t)pyr * desc_stride;
  const uint8_t *im = pyramids + (size_t)pyr * pyr_stride;
  const uint32_t img_bytes32 = (uint32_t)((size_t)P.rows * P.vstep);
  const int vstep = P.vstep;
  lds_u8 *wave_patches = (lds_u8 *)(patches + (wv * 2) * ORB_PATCH_BYTES);
  const lds_u32 *kpl_l = (const lds_u32 *)kpl, *kpos_l = (const lds_u32 *)kpos;
  // rounds of at most per_max keypoints (one round unless the pyramid holds more keypoints than max_keypoints)
  for (uint32_t c0 = lo; c0 < hi; c0 += per_max) {
    const uint32_t c1 = min(c0 + per_max, hi);
    if (tid == 0) sh->ntodo = 0;
    __syncthreads();
    // ---- one thread per staged keypoint: strip by binary search, final position, keypoint + descriptor ----
    for (uint32_t e = c0 + tid; e < c1; e += 256) {
      int a = 0, b = S;                                // largest strip index with soff[a] <= e
      while (b - a > 1) {
        const int m = (a + b) >> 1;
        if (soff[m] <= e) a = m; else b = m;
      }
      const uint32_t k = e - soff[a];
      const uint32_t ss = sslot[a];
      const uint32_t v = stage[(ss & 0x7fffffffu) + k];
      uint32_t pos = e;                                // (whole-level entries: storage order IS the reference's order)
      if (ss >> 31) {
        const PlanStrip ps = plan_strip(P, a);
        pos = final_position(P, soff, stage, ps.li, ps.s, k, v);
      }
      if (pos >= cap) continue;                        // beyond the caller's capacity: counted, not stored
      kp[(size_t)pyr * kp_stride + pos] = v;
      if (cnt[a] & STRIP_DESCRIBED) {
        const uint32_t *src = sd + ((size_t)a * QS_SHARED + k) * words;
        for (int w = 0; w < words; w++) dsc[(size_t)pos * words + w] = src[w];
      } else {
        const uint32_t slot = atomicAdd(&sh->ntodo, 1u);   // (order irrelevant: results are positional)
        kpl[slot] = v;
        kpos[slot] = pos;
      }
    }
    __syncthreads();
    const uint32_t nt = sh->ntodo;
    const int glevel = GHOOKS ? (P.ablate >> 20) & 15 : 0;
    if (nt != 0 && !(P.ablate & 64) && glevel != 1) {
      // ---- describe the rest: two keypoints per wave iteration ----
      const uint32_t npairs = (nt + 1) >> 1;
      // the two keypoints of pair `it`: packed words (0 when absent) and their final positions
      auto pair_of = [&](uint32_t it, uint32_t &p0, uint32_t &p1, uint32_t &q0, uint32_t &q1) {
        const bool h0 = it < npairs, h1 = it < npairs && 2 * it + 1 < nt;
        p0 = h0 ? kpl_l[2 * it] : 0u;
        q0 = h0 ? kpos_l[2 * it] : 0u;
        p1 = h1 ? kpl_l[2 * it + 1] : 0u;
        q1 = h1 ? kpos_l[2 * it + 1] : 0u;
      };
      // Two register sets (A, B) in ping-pong: the loads of the next pair are in flight while the current one
      // is described, without copying 12 registers per iteration.
      uint32_t a0, a1, b0, b1, qa0, qa1, qb0, qb1;
      pair_of(wv, a0, a1, qa0, qa1);
      OrbWin wa = orb_fetch(G, a0, a1, im, vstep, img_bytes32), wb;
      for (uint32_t it = wv; it < npairs; it += 2 * OWAVES) {
        pair_of(it + OWAVES, b0, b1, qb0, qb1);
        wb = orb_fetch(G, b0, b1, im, vstep, img_bytes32);
        orb_describe<lds_u8, GHOOKS>(G, wa, a0, a1, wave_patches, vstep, rtab, words, (uint8_t *)dsc, (G.half ? qa1 : qa0) * (uint32_t)words, glevel);
        if (it + OWAVES >= npairs) break;
        pair_of(it + 2 * OWAVES, a0, a1, qa0, qa1);
        wa = orb_fetch(G, a0, a1, im, vstep, img_bytes32);
        orb_describe<lds_u8, GHOOKS>(G, wb, b0, b1, wave_patches, vstep, rtab, words, (uint8_t *)dsc, (G.half ? qb1 : qb0) * (uint32_t)words, glevel);
      }
    }
    __syncthreads();                                  // kpl / kpos / todo are reused by the next round
  }
}

// ===========================================================================
// k_gather_orb — strip lists -> final keypoint order; descriptors copied from the strips' staging slots,
// or computed here for the strips that did not describe their own keypoints.
//
// grid (NCH, batch): workgroup (ch, pyr) owns the staged keypoints [ch*per, (ch+1)*per) of pyramid pyr in
// storage order.  Every workgroup redoes the (tiny) exclusive scan of the pyramid's strip counts, then, per
// keypoint (one thread each, its strip found by binary search): final position (final_position), keypoint
// written there, and
//   - the descriptor copied from the strip's descriptor slots when the strip described it (strip count
//     bit 31, see strip_body phase E);
//   - otherwise queued and described here: orb_fetch / orb_describe, two keypoints per wave iteration, the
//     next pair's loads in flight while the current pair is processed.
// ===========================================================================
// (waves_per_eu(6): asking the compiler for 7 waves caps its SGPRs and cost 13 SGPR spills — v_writelane / v_readlane in
//  the describe loop; the resident waves are pinned to 6 per SIMD inside the kernel, see there.  A third register set
//  (the loads of TWO pairs in flight) was measured too: 80 VGPRs, no gain.)
template <bool GHOOKS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6))) void k_gather_orb(
    const FusedParams P, const uint8_t *__restrict__ pyramids, size_t pyr_stride,
    const uint32_t *__restrict__ stage_kp, const uint32_t *__restrict__ strip_count,
    const uint32_t *__restrict__ stage_desc,
    uint32_t *__restrict__ kp, size_t kp_stride, uint32_t cap, uint32_t *__restrict__ counts,
    uint32_t *__restrict__ desc, size_t desc_stride, int words, uint32_t per_max, uint32_t *__restrict__ ovf_reset) {
  // SIX waves per SIMD, pinned: the clobber makes the allocation 80 VGPRs (the kernel uses 72).  With three batches in
  // flight this kernel shares the CUs with the next batch's strip kernel, and the step is fastest when it holds six
  // wave slots per SIMD: measured 0.2261 ms at 6, 0.2293 at 7 (what 72 VGPRs allow: the kernel ALONE is 2 % faster
  // there, its waves live 13 % longer), 0.2302 at 5, 0.2354 at 4.  Until round 4 the limit of 6 came about by accident
  // — 101 SGPRs — and went away whenever a change to the describe loop freed a few of them.
  asm volatile("" ::: "v79");
  extern __shared__ __attribute__((aligned(16))) uint8_t osm[];
  // the overflow list of this step has been consumed (stream order): empty it for the next step
  if (ovf_reset && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    ovf_reset[1] = ovf_reset[0];                     // kept for pislam_frontend_last_stats
    ovf_reset[0] = 0;
  }
  orb_role<NoWait, GHOOKS>(P, pyramids, pyr_stride, stage_kp, strip_count, stage_desc, kp, kp_stride, cap, counts, desc, desc_stride,
                           words, per_max, osm, (int)blockIdx.y, (int)blockIdx.x, (int)gridDim.x);
}

// ===========================================================================
// k_frame — the whole path in ONE launch for small batches (the reference's own use: a frame at a time, demo.cpp:77-101,
// README.md:17-20).  Three dependent launches cost three launch floors (~6.5 us each on MI355X, hipGraph replay
// included): one pyramid took 31 us end to end, 15 us of it work.  Here the grid holds the strip workgroups FIRST
// (run-major, longest runs first, like k_fused_strips) and the gather + ORB workgroups of every pyramid BEHIND them;
// an ORB workgroup waits until its pyramid's runs have all published their lists.
//   * Forward progress: workgroups are dispatched in blockIdx order, so whenever an ORB workgroup is resident every strip
//     workgroup has been dispatched or is about to be, and a strip workgroup never waits for anything; the host takes
//     this path only while the waiting workgroups of all such launches in flight are a small fraction of an XCD's resident
//     slots (run_fused).  No AMD document promises that order (CU masks, debuggers, future firmware), so the wait is
//     BOUNDED and a timeout is never silent: the workgroup that gives up raises a sticky flag in device memory and in a
//     host-mapped word the library reads at the start of every call, publishes counts[pyr] = PISLAM_COUNT_INVALID
//     (0xffffffff) instead of a count, and the launch does not re-arm its counters; every later launch sees the flag and
//     publishes the same sentinel until the host has reset the counters (and stopped taking this path on that context).
//   * Hand-over: a strip workgroup finishes with a workgroup barrier (which also drains its stores) and ONE agent-scope
//     RELEASE increment of its pyramid's counter (the L2 write-back of its XCD: pyramids are spread over all eight);
//     the ORB workgroup polls with relaxed agent-scope loads, then ONE acquire fence and a workgroup barrier.
//   * Strips whose queues overflow are redone in place (strips_role<INLINE_OVF>): no overflow list, no second launch.
//   * The last ORB workgroup to finish re-arms the counters for the next launch (and keeps the deferred-strip count for
//     pislam_frontend_last_stats) — a captured launch replays without any host-side reset.
// sync: [0 .. FRAME_SYNC_PYR) runs done per pyramid, [FRAME_SYNC_DONE] ORB workgroups done, [FRAME_SYNC_POISON] sticky:
// a wait timed out (only the host clears it).  `hflag`: the host-mapped copy of the sticky flag.
// `test`: bit 0 = the first strip workgroup skips its release (forces the timeout: tests), bits 8.. = log2 of the poll limit.
// ===========================================================================
constexpr int FRAME_SYNC_PYR = 8, FRAME_SYNC_DONE = 8, FRAME_SYNC_POISON = 9, FRAME_SYNC_WORDS = 12;
constexpr uint32_t COUNT_INVALID = 0xffffffffu;     // == PISLAM_COUNT_INVALID
__global__ __launch_bounds__(NT) void k_frame(const FusedParams P, const uint8_t *__restrict__ pyramids, size_t pyr_stride,
                                              uint32_t *__restrict__ stage_kp, uint32_t *__restrict__ strip_count,
                                              uint32_t *__restrict__ kp, size_t kp_stride, uint32_t cap,
                                              uint32_t *__restrict__ counts, uint32_t *__restrict__ desc, size_t desc_stride,
                                              int words, uint32_t per_max, int nch, uint32_t *__restrict__ sync,
                                              uint32_t *__restrict__ ovf, uint32_t *__restrict__ hflag, uint32_t test) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ uint32_t sh_ctr[16];
  const int n_strip = P.batch * P.runs_per_pyr;
  if ((int)blockIdx.x < n_strip) {
    const int pyr = (int)blockIdx.x % P.batch, run = (int)blockIdx.x / P.batch;
    strips_role<true, false, true, false, false, true>(P, pyramids, pyr_stride, stage_kp, strip_count, nullptr, 0, nullptr, ovf,
                                                       nullptr, smem, sh_ctr, pyr, run);
    __syncthreads();                                // every thread's stores have been issued and acknowledged
    if (threadIdx.x == 0 && !((test & 1u) && blockIdx.x == 0))
      (void)__hip_atomic_fetch_add(&sync[pyr], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const int ob = (int)blockIdx.x - n_strip, pyr = ob / nch, ch = ob - pyr * nch;
  __shared__ uint32_t sh_ok;
  auto wait = [&]() -> bool {
    if (threadIdx.x == 0) {
      uint32_t ok = 0;
      // an earlier launch timed out and the host has not reset the counters yet: they cannot be trusted
      const bool poisoned = __hip_atomic_load(&sync[FRAME_SYNC_POISON], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
      // (relaxed polls — an acquire load would invalidate this XCD's caches on every iteration, under the strip workgroups
      //  still streaming their rows through them — and ONE acquire fence once the counter is there)
      const int limit = poisoned ? 0 : 1 << (((test >> 8) & 31u) ? ((test >> 8) & 31u) : 20u);   // (2^20 polls ~ 1 s: only a lost launch ever gets there)
      for (int it = 0; it < limit; it++) {
        if (__hip_atomic_load(&sync[pyr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (uint32_t)P.runs_per_pyr) {
          ok = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(PISLAM_FRAME_SLEEP);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      if (!ok) {
        __hip_atomic_store(&sync[FRAME_SYNC_POISON], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(hflag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (ch == 0) counts[pyr] = COUNT_INVALID;   // never a stale count: the caller's own data says the call failed
      }
      sh_ok = ok;
    }
    __syncthreads();
    return sh_ok != 0;
  };
  orb_role(P, pyramids, pyr_stride, stage_kp, strip_count, nullptr, kp, kp_stride, cap, counts, desc, desc_stride, words, per_max,
           smem, pyr, ch, nch, wait);
  __syncthreads();
  // (a workgroup that gave up does not count as done: a launch that timed out never re-arms — late strip workgroups may
  //  still be incrementing its counters; the host resets them)
  if (threadIdx.x == 0 && sh_ok != 0) {
    const uint32_t done = __hip_atomic_fetch_add(&sync[FRAME_SYNC_DONE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done + 1 == (uint32_t)(P.batch * nch)) {    // the launch's last workgroup: every strip and every ORB workgroup is through
      for (int i = 0; i <= FRAME_SYNC_DONE; i++) __hip_atomic_store(&sync[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ovf[1] = __hip_atomic_load(&ovf[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // strips redone in place (last_stats)
      __hip_atomic_store(&ovf[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace pf
