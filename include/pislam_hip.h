/*
 * pislam_hip.h — C ABI of the MI355X-native PiSlam ORB front-end.
 *
 * This is the drop-in boundary: a plain-C shared library (libpislam_hip.so,
 * built from pislam_amd/csrc with hipcc for gfx950) whose entry points are
 * what the reference's header-only templates forward to.  The C++ headers in
 * include/pislam/ (Fast.h, Harris.h, Orb.h, Brief.h, Util.h) keep the
 * reference's names, template parameter lists and argument order and are thin
 * wrappers over the functions below; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every function returns PISLAM_OK (0) or a negative PISLAM_ERR_* code;
 *     pislam_last_error(ctx) returns a static/ctx-owned message.
 *   - image / score-map pointers are row-major uint8 with a row stride of
 *     `vstep` bytes, exactly the reference's `uint8_t img[][vstep]`.
 *   - every data pointer may be a HOST pointer or a DEVICE pointer
 *     (hipPointerGetAttributes decides).  Host data is staged through
 *     ctx-owned device buffers; device data is used in place (zero copy).
 *   - all work is issued on the ctx's stream (default: the null stream);
 *     calls with host pointers synchronise before returning, calls with only
 *     device pointers are asynchronous on that stream unless noted.
 *   - there is NO CPU fallback: if no gfx950 device is usable the functions
 *     return PISLAM_ERR_HIP.
 */
#ifndef PISLAM_HIP_H_
#define PISLAM_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PISLAM_OK 0
#define PISLAM_ERR_INVALID (-1) /* bad argument / violated precondition */
#define PISLAM_ERR_HIP (-2)     /* HIP runtime error (no device, launch failure, ...) */
#define PISLAM_ERR_NOMEM (-3)   /* device allocation failed */
#define PISLAM_ERR_DIST (-4)    /* RCCL unavailable or a collective failed */

/* 2: pislam_pyramid_build_batch takes a PISLAM_BUILD_* bitmask (ABI 1: `blur`, any non-zero value) and refuses
 *    unknown bits; pislam_dist_comm_count; option "own_stream" 1 creates a stream with DEFAULT flags (ordered
 *    with the legacy null stream), 2 a non-blocking one; options "sub_batches" / "sub_mb". */
#define PISLAM_ABI_VERSION 2

typedef struct pislam_ctx pislam_ctx;

/* ---- context ---------------------------------------------------------- */
int pislam_abi_version(void);
/* device < 0 selects the current HIP device. */
int pislam_ctx_create(int device, pislam_ctx **ctx);
int pislam_ctx_destroy(pislam_ctx *ctx);
/* hip_stream is a hipStream_t passed as void*; NULL = null stream. */
int pislam_ctx_set_stream(pislam_ctx *ctx, void *hip_stream);
/* Tuning / test hooks; results never depend on them.  Keys:
 *   "own_stream" issue on a stream created (and destroyed) by the context instead of the null stream: 1 = default flags
 *                (still ordered with work on the legacy null stream, like the null stream itself), 2 = hipStreamNonBlocking
 *                (device-pointer inputs must then be complete, or ordered by the caller, before a call); 0 = back to the null stream
 *   "sub_batches" fused batch path: 1 (default) one launch group; n <= 16: the batch is cut into n sub-batches whose overflow
 *                pass + gather/ORB kernels run on a context-owned second stream under the next sub-batch's strip kernel
 *                (fork / join by events inside the call; results and stream semantics unchanged, hipGraph-capturable;
 *                measured slower than one launch group on MI355X — DESIGN.md); 0: by size ("sub_mb" MiB per sub-batch, 128)
 *   "pipeline"   0 auto, 1 staged (one launch group per reference call, HBM score map), 2 fused strips
 *   "dump_score" fused pipeline also materialises the score map (pislam_frontend_get_score_map)
 *   "strip_rows" fused strip height (0 = heuristic);  "run_len" strips per workgroup run (0 = by batch)
 *   "alias"      1 (default) score tile laid over the dead image rows + overflow pass, 0 separate tiles
 *   "orb_chunks" gather+ORB workgroups per pyramid
 *   "run_order"  1 (default) a pyramid's runs are launched longest first, 0 in level order
 *   "tile_cols"  levels with more classified columns are cut into x-tiles run by separate workgroups (0 = 704, < 0 never)
 *   "orb_in_strip" 1 strips describe their own keypoints right after their NMS, 0 (default) one gather+ORB pass describes all
 *   "bucket_select" fused batch path with log_bucket_size != 0: 1 (default) the strips run as without buckets and a selection
 *                pass between strip kernel and gather keeps each bucket's bucket_limit largest keypoints (log_bucket_size 1..8);
 *                0 the selection happens inside the strips (strips cut on bucket rows; log_bucket_size 2..5, others take the
 *                staged pipeline)
 *   "frame"      small batches run as ONE launch (strip workgroups, then the gather + ORB workgroups of the same grid behind an
 *                agent-scope hand-over; overflowed strips redone in place): 1 (default) batches of 1 or 2 pyramids, n = 2..8
 *                batches of up to n, 0 never (always strip kernel -> overflow pass -> gather + ORB); not with buckets
 *   "wgs_per_cu", "strip_px", "strip_rows_max", "lds_pad", "bucket_round_up", "repeat_strips", "ablate"  profiling only (ablate != 0 gives INVALID results by design)
 *   "match_mfma" 1 (default) pislam_match_hamming* run on the int8 matrix cores, 0 the VALU popcount kernel (same results)
 *   "dist_rccl_single" test hook: pislam_dist_init(world = 1) still creates a 1-rank RCCL communicator */
int pislam_ctx_set_option(pislam_ctx *ctx, const char *key, int value);
int pislam_ctx_synchronize(pislam_ctx *ctx);
const char *pislam_last_error(const pislam_ctx *ctx);

/* ---- the four reference entry points (one pyramid level each) ---------- */

/* replaces pislam::fastDetect<vstep,border>(width,height,img,out,threshold)
 * — reference include/Fast.h:54-158.  FAST-9 segment test; writes 0xff/0x00
 * to out rows [border,height-border), columns [border, border+16*ceil((width-
 * 2*border)/16)), plus out[y][width]=out[y][width+1]=0 when width%16 != 0.
 * Nothing else in `out` is touched.  Requires border >= 3.  Like the
 * reference's 16-byte vectors, the over-classified columns are addressed
 * flat: when border + 16*ceil((width-2*border)/16) + 3 > vstep they continue
 * in the next row, and `img` must be readable up to byte
 * (height-border+2)*vstep + that column (a few bytes past height*vstep only
 * for border < 6 on a level as wide as vstep). */
int pislam_fast_detect(pislam_ctx *ctx, int vstep, int border, int width, int height,
                       const uint8_t *img, uint8_t *out, int threshold);

/* replaces pislam::fastScoreHarris<vstep,border>(width,height,img,threshold,out)
 * — reference include/Fast.h:166-180 (+ Harris.h:37-248).  Every non-zero
 * out[y][x], y in [border,height-border), x in [border,width-border) is
 * replaced by its 8-bit Harris log-score.  Requires border >= 4. */
int pislam_fast_score_harris(pislam_ctx *ctx, int vstep, int border, int width, int height,
                             const uint8_t *img, int32_t threshold, uint8_t *out);

/* replaces pislam::fastExtract<vstep,border,logBucketSize,bucketLimit>(width,
 * height,out,results) — reference include/Fast.h:196-355.  2x2-block non-max
 * suppression (+ optional per-bucket top-k).  Writes the packed keypoints
 * (score<<24 | x<<12 | y, level-relative, reference order) to results[0..cap)
 * and the number the reference would have appended to *count (it may exceed
 * `capacity`; only the first `capacity` are stored).  logBucketSize 0..8,
 * bucketLimit 1..64.  Synchronous (count is returned to the host). */
int pislam_fast_extract(pislam_ctx *ctx, int vstep, int border, int logBucketSize,
                        int bucketLimit, int width, int height, const uint8_t *out,
                        uint32_t *results, size_t capacity, size_t *count);

/* replaces pislam::orbCompute<vstep,words>(img,points,descriptors)
 * — reference include/Orb.h:396-441.  descriptors[i*words + j], i in [0,n).
 * `img` is the base of the (stacked) image the keypoint coordinates refer to;
 * with a host pointer only the byte hull the reference itself reads
 * (rows y-15..y+15, columns x-15..x+16 of every keypoint) is accessed.
 * words 1..8. */
int pislam_orb_compute(pislam_ctx *ctx, int vstep, int words, const uint8_t *img,
                       const uint32_t *points, size_t n, uint32_t *descriptors);

/* ---- the reference's public helpers (L1 primitives) -------------------- */

/* pislam::harrisScoreSobel<vstep>(img,x,y,threshold) — Harris.h:80-248.
 * Batched over n points; scores[i] for packed point i (x<<12|y, score ignored). */
int pislam_harris_score_points(pislam_ctx *ctx, int vstep, const uint8_t *img,
                               const uint32_t *points, size_t n, int32_t threshold,
                               uint8_t *scores);

/* pislam::orbCentroids<vstep>(img,points) — Orb.h:80-308.  centroids has
 * pislam_centroids_size(n) int32 in the reference's grouped layout
 * [x0 x1 x2 x3 y0 y1 y2 y3]..., padding slots zero. */
size_t pislam_centroids_size(size_t n);
int pislam_orb_centroids(pislam_ctx *ctx, int vstep, const uint8_t *img,
                         const uint32_t *points, size_t n, int32_t *centroids);

/* pislam::atan2(const std::vector<int32_t>&) — Orb.h:310-387.  n8 (multiple
 * of 8) int32 in the grouped layout -> n8/2 angle bins (0..29), padding slots
 * included, exactly like the reference. */
int pislam_orb_angles(pislam_ctx *ctx, const int32_t *centroids, size_t n8, uint8_t *angles);

/* pislam::briefDescribe<vstep,words>(img,x,y,rot,descriptor) — Brief.h:637-733,
 * batched: descriptors[i*words+j] for point i with rotation rots[i]. */
int pislam_brief_describe(pislam_ctx *ctx, int vstep, int words, const uint8_t *img,
                          const uint32_t *points, const uint8_t *rots, size_t n,
                          uint32_t *descriptors);

/* The rotated-BRIEF offset table int8[30][256][4] = (dx0,dy0,dx1,dy1) the
 * kernels use (behaviour of Brief.h:28-53); host memory, 30720 bytes. */
const int8_t *pislam_brief_table(void);

/* ---- image preparation ("next" tier) -------------------- */

/* replaces pislam::gaussian5x5<vstep>(width,height,img,out) — reference include/Gaussian.h:48.
 * Separable [1 4 6 4 1]/16 built from rounding halving adds, vertical then horizontal, reflect-101
 * borders: bit-exact to the reference's stated expectation (test/GaussianTest.cpp:159-215).
 * Writes the width x height region only; img == out (in place) is allowed. */
int pislam_gaussian5x5(pislam_ctx *ctx, int vstep, int width, int height, const uint8_t *img,
                       uint8_t *out);

/* replace pislam::bilinear7_8<vstep> / bilinear13_16<vstep>(width,height,img,out) — reference
 * include/Bilinear.h:42 / :165; arithmetic of test/BilinearTest.cpp:171-196 / :198-233.  The image
 * must be padded to a multiple of 8 / 16 in both dimensions (Bilinear.h:32,155); output dimensions
 * round down (floor(w*7/8) x floor(h*7/8), resp. 13/16); like the reference, whole 7x7 / 13x13
 * output blocks are written.  img == out is allowed. */
int pislam_bilinear7_8(pislam_ctx *ctx, int vstep, int width, int height, const uint8_t *img,
                       uint8_t *out);
int pislam_bilinear13_16(pislam_ctx *ctx, int vstep, int width, int height, const uint8_t *img,
                         uint8_t *out);

/* ---- the measured path: a batch of stacked pyramids, device resident --- */

typedef struct pislam_level {
  int32_t width;  /* level width  (pixels)                                   */
  int32_t height; /* level height (rows)                                     */
  int32_t row0;   /* first row of the level inside the stacked pyramid       */
  int32_t col0;   /* first column (0 for the vertically stacked layout)      */
} pislam_level;

typedef struct pislam_frontend_params {
  int32_t vstep;            /* row stride in bytes (reference template vstep)  */
  int32_t rows;             /* rows per pyramid buffer                         */
  int32_t nlevels;          /* 1..16                                           */
  int32_t border;           /* reference template border (>= 16 for ORB)       */
  int32_t fast_threshold;   /* fastDetect threshold (demo: 20)                 */
  int32_t harris_threshold; /* fastScoreHarris threshold (demo: 1<<15)         */
  int32_t log_bucket_size;  /* fastExtract logBucketSize (0 = no buckets)      */
  int32_t bucket_limit;     /* fastExtract bucketLimit                         */
  int32_t words;            /* orbCompute words (1..8)                         */
  int32_t max_keypoints;    /* capacity per pyramid of keypoints/descriptors   */
} pislam_frontend_params;

/* On-GPU pyramid build (BASELINE config 5).  The reference ships no pyramid builder (README.md:28-31)
 * but provides the two reductions to build one from (Bilinear.h:28-30,153): level k+1 =
 * bilinear7_8 (steps[k] = 1) or bilinear13_16 (steps[k] = 2) of level k, level 0 = gaussian5x5 of the
 * frame (blur != 0) or the frame itself.  pislam_pyramid_layout computes the level table of a vertically
 * stacked pyramid (dimensions round down, every level's slot keeps the padding rows its reduction
 * reads); pislam_pyramid_build_batch fills `batch` pyramids from `batch` device-resident frames.
 * Every byte a consumer reads — the levels, the whole output blocks of each reduction, the next reduction's
 * block padding and FAST's right-edge columns (a margin of 32 columns / 16 rows around each level's
 * rewritten rectangle, zeroed on every call) — equals running the reference functions level by level on a
 * zero-initialised buffer; bytes beyond those margins are left untouched. */
int pislam_pyramid_layout(int width, int height, int nlevels, const int32_t *steps, int vstep_min,
                          pislam_level *levels, int32_t *vstep, int32_t *rows);
#define PISLAM_BUILD_BLUR 1          /* level 0 = gaussian5x5 of the frame (else the frame itself)              */
#define PISLAM_BUILD_MARGINS_CLEAN 2 /* the margins are already zero: this function filled `pyramids` before with */
                                     /* the same layout and nothing else wrote to it since — skip re-zeroing them */
#define PISLAM_BUILD_CHECK_MARGINS 4 /* debug, with MARGINS_CLEAN: verify that promise (synchronises; non-zero    */
                                     /* margin bytes -> PISLAM_ERR_INVALID).  Other bits are refused.             */
int pislam_pyramid_build_batch(pislam_ctx *ctx, int nlevels, const int32_t *steps, const pislam_level *levels,
                               const uint8_t *frames, int frame_vstep, size_t frame_stride, int batch,
                               uint8_t *pyramids, int vstep, int rows, size_t pyramid_stride, int flags);

/* Runs, for every pyramid b in [0,batch) and every level l (in order), the
 * call sequence of reference demo/demo.cpp:77-101 / README.md:67-82:
 *   fastDetect -> fastScoreHarris -> fastExtract (y += row0, x += col0)
 * then one orbCompute over the stacked image, entirely on the device.
 *   pyramids    : DEVICE, batch * pyramid_stride bytes, pyramid b at
 *                 pyramids + b*pyramid_stride, uint8 [rows][vstep]
 *   keypoints   : DEVICE uint32 [batch][max_keypoints]   (reference order)
 *   descriptors : DEVICE uint32 [batch][max_keypoints][words]
 *   counts      : DEVICE uint32 [batch]  = keypoints the reference would emit
 *                 (entries beyond max_keypoints are dropped, count is not clamped)
 * Asynchronous on the ctx stream; no host round trips; capturable in a hipGraph
 * after pislam_frontend_reserve() has sized the workspace. */
int pislam_orb_frontend_batch(pislam_ctx *ctx, const pislam_frontend_params *params,
                              const pislam_level *levels, const uint8_t *pyramids,
                              size_t pyramid_stride, int batch, uint32_t *keypoints,
                              uint32_t *descriptors, uint32_t *counts);

/* Pre-allocates the ctx workspace for the given shape (optional; the batch
 * call grows it on demand, which synchronises). */
int pislam_frontend_reserve(pislam_ctx *ctx, const pislam_frontend_params *params,
                            const pislam_level *levels, int batch);

/* Debug / parity hook: after pislam_orb_frontend_batch, copies the internal
 * score map of pyramid b (uint8 [rows][vstep], what the reference's `out`
 * holds after fastScoreHarris on every level) to dst (host or device).
 * Returns PISLAM_ERR_INVALID if the active pipeline does not materialise it. */
int pislam_frontend_get_score_map(pislam_ctx *ctx, int b, uint8_t *dst);

/* Elapsed milliseconds of the LAST pislam_orb_frontend_batch call on this ctx,
 * measured with hipEvents recorded on the ctx stream around its kernels
 * (total, and per internal stage — staged pipeline: 0 detect+score, 1 extract,
 * 2 orb; fused pipeline: 0 strip kernel (detect+score+extract), 1 overflow
 * pass, 2 gather+orb).
 * Synchronises on the end event. */
int pislam_frontend_last_timing(pislam_ctx *ctx, float *total_ms, float stage_ms[3]);

/* Diagnostics of the last pislam_orb_frontend_batch call on the fused pipeline:
 * stats[0] = strips whose on-chip queues overflowed (very dense corners) and
 * that were redone by the slower overflow pass, stats[1] = strips in the call.
 * Results are identical either way; a large ratio means the input is denser
 * than the fast path is sized for.  Synchronises the context stream. */
int pislam_frontend_last_stats(pislam_ctx *ctx, uint32_t stats[2]);

/* Which path the last pislam_orb_frontend_batch call on this context took (a bit mask; 0 before the first call).
 * Results are identical on every path; this is for tests, tuning and bug reports.  No synchronisation. */
#define PISLAM_PATH_STAGED 1u          /* one launch group per level (option "pipeline" 1, or a shape the strips cannot take) */
#define PISLAM_PATH_FUSED 2u           /* strip kernel -> overflow pass -> gather + ORB */
#define PISLAM_PATH_ONE_LAUNCH 4u      /* the same work as ONE launch (pf::k_frame): batches of 1 or 2 pyramids by default, option "frame" */
#define PISLAM_PATH_BUCKET_SELECT 8u   /* buckets applied by the selection pass between strips and gather (option "bucket_select" 1) */
#define PISLAM_PATH_BUCKETS_IN_STRIPS 16u /* buckets applied inside the strips (option "bucket_select" 0, or more buckets than the pass holds) */
#define PISLAM_PATH_GENERIC_ORB 32u    /* generic gather + per-keypoint ORB kernels (vstep % 16 != 0) */
#define PISLAM_PATH_FRAME_TIMED_OUT 64u /* an earlier one-launch call of this context timed out (reported then, see below): the
                                          context runs small batches as three launches now */
unsigned pislam_frontend_last_path(const pislam_ctx *ctx);
/* (On a pipeline lane the value is that of the lane's last call whether it ran eagerly or was replayed from its
 * hipGraph.  On PISLAM_PATH_ONE_LAUNCH calls pislam_frontend_last_timing reports the whole call as stage 0: stages 1, 2 = 0.)
 *
 * The one-launch path's safety net.  Inside pf::k_frame the gather + ORB workgroups wait for the strip workgroups of the
 * same grid; the library takes the path only while such waiting workgroups are a small fraction of the resident slots,
 * and the wait is bounded (~1 s).  Should it ever expire (nothing observed does this: CU masks, a debugger or a future
 * dispatcher could), the call does NOT return stale data silently:
 *   - counts[i] of every pyramid whose keypoints were not produced is PISLAM_COUNT_INVALID (no valid count can be);
 *   - the next call on the context (or on its pipeline lane), pislam_ctx_synchronize, pislam_pipeline_synchronize and
 *     pislam_frontend_last_stats return PISLAM_ERR_HIP once, with a message naming the time-out, after resetting the
 *     hand-over state; from then on the context runs small batches as three launches (PISLAM_PATH_FRAME_TIMED_OUT is set
 *     in pislam_frontend_last_path; graphs the pipeline captured with a one-launch node are dropped). */
#define PISLAM_COUNT_INVALID 0xffffffffu

/* Host only — no device, no allocation, no launch: build the strip plan (and the bucket selection plan) a batch call with
 * these parameters would run and check the invariants the kernels rely on.  `options`: "key=value,key=value" with
 * pislam_ctx_set_option keys (NULL / "": defaults); num_cus <= 0: 256; lanes_in_flight: 1, or the depth of the pipeline the
 * call would be a lane of.  summary: [0] plan entries, [1] strips, [2] runs, [3] staging slots per pyramid, [4] strips per
 * run, [5] / [6] LDS bytes of the plain / aliased layout, [7] units of the bucket selection pass.  Returns
 * PISLAM_ERR_INVALID (message in `err`) when the parameters are refused or the staged pipeline would take the call. */
int pislam_debug_build_plan(const pislam_frontend_params *params, const pislam_level *levels, int batch, int num_cus,
                            int lanes_in_flight, const char *options, uint32_t summary[8], char *err, size_t err_cap);

/* ---- batches in flight ---------------------------------------------------
 * A pipeline = `depth` (1..8) contexts behind one object, each with its own workspace and non-blocking stream:
 * batch k runs on lane k % depth, so that the tail of one batch (partly filled CUs, the latency-bound gather+ORB
 * kernel, launch gaps) runs under the head of the next.  MI355X, 256 VGA pyramids per batch: 0.27 ms per batch
 * one call at a time, 0.23 ms with depth 3.  The reference loop (demo/demo.cpp:77-101: frames are independent)
 * becomes
 *     for each batch k:  pislam_pipeline_submit(pipe, ..., inputs_k, outputs_k, producer_stream, 1, &t[k]);
 *     before consuming outputs_k on stream s:  pislam_pipeline_wait(pipe, t[k], s);
 * submit: the lane's stream first waits (device side) for everything `input_stream` (hipStream_t as void*) holds
 * so far when order_after_input != 0 — the producer of `pyramids` — then runs pislam_orb_frontend_batch.  Outputs
 * of a batch must stay untouched until its ticket has been waited for; a lane's batches are ordered among
 * themselves.  pislam_pipeline_stream gives the lane's stream of a ticket (e.g. for
 * pislam_dist_allgather_counts_on), pislam_pipeline_lane its context (statistics).  A call that repeats exactly
 * (same buffers, same shape: a steady stream of batches) is replayed from a hipGraph from its third occurrence
 * on (option "graphs" 0 = always eager); pislam_pipeline_set_option applies any context option to every lane. */
typedef struct pislam_pipeline pislam_pipeline;
int pislam_pipeline_create(int device, int depth, pislam_pipeline **pipe);
int pislam_pipeline_destroy(pislam_pipeline *pipe);
int pislam_pipeline_depth(const pislam_pipeline *pipe);
int pislam_pipeline_set_option(pislam_pipeline *pipe, const char *key, int value);
int pislam_pipeline_reserve(pislam_pipeline *pipe, const pislam_frontend_params *params, const pislam_level *levels,
                            int batch);
int pislam_pipeline_submit(pislam_pipeline *pipe, const pislam_frontend_params *params, const pislam_level *levels,
                           const uint8_t *pyramids, size_t pyramid_stride, int batch, uint32_t *keypoints,
                           uint32_t *descriptors, uint32_t *counts, void *input_stream, int order_after_input,
                           uint64_t *ticket);
int pislam_pipeline_wait(pislam_pipeline *pipe, uint64_t ticket, void *stream);
int pislam_pipeline_synchronize(pislam_pipeline *pipe);
/* stats[0] batches submitted, [1] of them replayed from a hipGraph, [2] calls captured, [3] captures that failed
 * (such a call stays eager). */
int pislam_pipeline_stats(const pislam_pipeline *pipe, uint64_t stats[4]);
void *pislam_pipeline_stream(pislam_pipeline *pipe, uint64_t ticket);
pislam_ctx *pislam_pipeline_lane(pislam_pipeline *pipe, int lane);
const char *pislam_pipeline_last_error(const pislam_pipeline *pipe);

/* Measurement aid: the shader clock in GHz while the device is doing whatever else it is doing — one wave on the
 * context stream compares the shader cycle counter (s_memtime) with the constant 100 MHz counter
 * (s_memrealtime) over ~`micros` microseconds.  Synchronises.  bench.py prices the VALU issue rate with it. */
int pislam_debug_shader_clock(pislam_ctx *ctx, int micros, double *ghz);

/* ---- descriptor matching ---------------------------- */

/* The reference ships no matcher (README.md:125-128 only names matching as
 * the consumer of these descriptors), so there is no reference interface to
 * mirror: the semantics below are this library's own.
 *
 * Brute-force Hamming matching of `words`-dword binary descriptors
 * (words in {1,2,4,8}, as orbCompute produces them — Orb.h:396).  For every
 * query i: idx[i] = the train index with the smallest Hamming distance (ties:
 * the smallest index; -1 if nt == 0), dist[i] = that distance (0xffffffff if
 * nt == 0), dist2[i] = the smallest distance among all OTHER train descriptors
 * (0xffffffff if nt < 2; for a ratio test).  nt <= 65535.  Host or device
 * pointers. */
int pislam_match_hamming(pislam_ctx *ctx, int words, const uint32_t *query, size_t nq,
                         const uint32_t *train, size_t nt, int32_t *idx, uint32_t *dist,
                         uint32_t *dist2);

/* Batched form on device-resident front-end outputs: pair b matches the first
 * min(qcounts[b], q_stride) descriptors of query[b] against the first
 * min(tcounts[b], t_stride) of train[b] (layouts [batch][stride][words], the
 * descriptor / count arrays of pislam_orb_frontend_batch; strides in
 * descriptors, t_stride <= 65535).  Outputs are [batch][q_stride]; entries
 * at and beyond the pair's query count are not written.  Device pointers
 * only; asynchronous on the context stream. */
int pislam_match_hamming_batch(pislam_ctx *ctx, int words, const uint32_t *query,
                               const uint32_t *qcounts, size_t q_stride, const uint32_t *train,
                               const uint32_t *tcounts, size_t t_stride, int batch, int32_t *idx,
                               uint32_t *dist, uint32_t *dist2);

/* ---- multi-GPU: one process per GPU, pyramids sharded, ONE collective ---- */

/* The reference is a single-threaded per-frame loop without cross-frame state
 * (demo/demo.cpp:77-101, README.md:59-82), so a batch of pyramids shards across
 * GPUs with no data-path exchange: rank r of `world` runs
 * pislam_orb_frontend_batch on its own contiguous range of pyramids.  The only
 * exchange is the all-gather of the per-pyramid keypoint counts (what a consumer
 * needs to place every rank's keypoints in one global list) — ncclAllGather over
 * RCCL/xGMI, 4 bytes per pyramid.  RCCL is bound at run time (librccl.so.1;
 * override with the environment variable PISLAM_RCCL_LIB); world == 1 never
 * touches it.  PISLAM_DIST_TRACE=1 makes pislam_dist_finalize print the host time
 * spent in each call of an exchange (a diagnostic, stderr).
 *
 *   rank 0:   pislam_dist_get_unique_id(id);  -> hand `id` to every rank (file, socket, MPI, env ...)
 *   all:      pislam_ctx_create(local_device, &ctx);  pislam_dist_init(ctx, id, rank, world);
 *   per step: pislam_orb_frontend_batch(ctx, ..., counts_local);
 *             pislam_dist_allgather_counts(ctx, counts_local, n_local, counts_all);
 *   end:      pislam_dist_synchronize(ctx);  pislam_dist_finalize(ctx);
 */
#define PISLAM_DIST_ID_BYTES 128

/* Contiguous split of `global_batch` pyramids: the first (global_batch % world)
 * ranks take one more.  Pure arithmetic; no context, no GPU. */
int pislam_dist_shard(int global_batch, int rank, int world, int *first, int *count);

/* ncclGetUniqueId: call on ONE rank, distribute the bytes to all of them. */
int pislam_dist_get_unique_id(uint8_t id[PISLAM_DIST_ID_BYTES]);

/* ncclCommInitRank on the context's device (collective: every rank must call
 * it with the same id and world).  Creates the context's collective stream.
 * world == 1: no communicator, `id` may be NULL. */
int pislam_dist_init(pislam_ctx *ctx, const uint8_t id[PISLAM_DIST_ID_BYTES], int rank, int world);
int pislam_dist_rank(const pislam_ctx *ctx);
int pislam_dist_world(const pislam_ctx *ctx);
/* Ranks in the context's communicator as RCCL itself reports them (ncclCommCount) — not the `world` the caller
 * passed in: 0 without a communicator (single GPU / not initialised), negative on error. */
int pislam_dist_comm_count(pislam_ctx *ctx);

/* all_counts[r*n + i] = rank r's local_counts[i] (DEVICE pointers, n equal on
 * every rank — pad ragged shards to the largest).  Enqueued on the context's
 * COLLECTIVE stream, ordered after everything enqueued on the context stream so
 * far; asynchronous to the host and to the context stream, so the next batch
 * call overlaps it.  local_counts / all_counts must stay untouched until the
 * collective has completed (pislam_dist_fence / pislam_dist_synchronize). */
int pislam_dist_allgather_counts(pislam_ctx *ctx, const uint32_t *local_counts, size_t n,
                                 uint32_t *all_counts);

/* Makes the context stream wait (on the device, not the host) for the
 * collective issued `back` calls ago (1 = the most recent): call it before
 * work that overwrites that collective's buffers.  With two alternating output
 * sets, pislam_dist_fence(ctx, 2) before each batch call is enough.  (The
 * completion events of the last 16 collectives are kept; an older one is
 * covered by waiting for the oldest kept — the collective stream is in order.) */
int pislam_dist_fence(pislam_ctx *ctx, int back);

/* The same two calls for a process that runs SEVERAL contexts / streams (batches in flight on separate
 * pipelines) over ONE communicator: the all-gather is ordered after `producer_stream` (a hipStream_t as void*)
 * instead of the context stream, the fence makes `consumer_stream` wait.  All collectives of the process then
 * go through one communicator and one collective stream, in host issue order — identical on every rank. */
int pislam_dist_allgather_counts_on(pislam_ctx *ctx, void *producer_stream, const uint32_t *local_counts, size_t n,
                                    uint32_t *all_counts);
int pislam_dist_fence_on(pislam_ctx *ctx, int back, void *consumer_stream);

/* Blocks the host until every collective issued on this context has completed. */
int pislam_dist_synchronize(pislam_ctx *ctx);

/* MAX of one host double over all ranks (ncclAllReduce; blocking; also a
 * barrier) — the "slowest rank" reduction of a timed region. */
int pislam_dist_allreduce_max(pislam_ctx *ctx, double *value);

/* Destroys the communicator and the collective stream (pislam_ctx_destroy does
 * this too). */
int pislam_dist_finalize(pislam_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* PISLAM_HIP_H_ */
