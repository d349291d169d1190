// pislam_hip.hip — C ABI (include/pislam_hip.h) over the gfx950 kernels.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC
// There is no CPU fallback anywhere in this library.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <string>
#include <vector>

#include "../../include/pislam_hip.h"

// rotated BRIEF table (behaviour of reference Brief.h:28-53), packed per pair
__device__ const uint32_t g_brief_tab[30 * 256] = {
#include "brief_table.inc"
};
static constexpr uint32_t h_brief_tab_packed[30 * 256] = {
#include "brief_table.inc"
};
// The same table as byte offsets into the LDS patch of the ORB kernels (rows pf::orb_row_ofs apart) whose row 15 /
// column 15 is the keypoint: ofs = row_ofs(dy+15) + (dx+15); low half = first
// sample point, high half = second.  Entry order: see make_brief_ofs.
struct BriefOfsTab {
  uint32_t v[30 * 256];
};
#include "pislam_dev.h"
__device__ const pdev::VrecpeTab g_vrecpe_tab = pdev::make_vrecpe_tab();   // used by pf::k_gather_orb
__device__ const pdev::OrbMaskTab g_orb_masks = pdev::make_orb_mask_tab();  // used by pf::orb_lane
extern __device__ const BriefOfsTab g_brief_ofs;
#include "pislam_stage_kernels.h"
#include "pislam_fused_kernels.h"
static constexpr int brief_row_ofs(int r) { return pf::orb_row_ofs(r); }
static constexpr BriefOfsTab make_brief_ofs() {
  BriefOfsTab t{};
  for (int i = 0; i < 30 * 256; i++) {
    const uint32_t e = h_brief_tab_packed[i];
    const int dx0 = (int8_t)(e & 0xff), dy0 = (int8_t)((e >> 8) & 0xff);
    const int dx1 = (int8_t)((e >> 16) & 0xff), dy1 = (int8_t)(e >> 24);
    // stored as [rot][t][r] for test k = 8 r + t (pf::orb_describe: lane r of a half runs tests 8 r .. 8 r + 7)
    const int rot = i >> 8, k = i & 255;
    t.v[rot * 256 + 32 * (k & 7) + (k >> 3)] =
        (uint32_t)(brief_row_ofs(dy0 + 15) + dx0 + 15) | ((uint32_t)(brief_row_ofs(dy1 + 15) + dx1 + 15) << 16);
  }
  return t;
}
__device__ const BriefOfsTab g_brief_ofs = make_brief_ofs();

#include "pislam_prep_kernels.h"
#include "pislam_match_kernels.h"

#define PISLAM_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  unsigned long long reallocs = 0;   // times the buffer moved: a captured hipGraph holds the OLD address
  // returns true if (re)allocated
  int ensure(size_t bytes, bool *grew = nullptr) {
    if (grew) *grew = false;
    if (bytes <= cap) return PISLAM_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = std::max<size_t>(bytes, 256);
    if (hipMalloc(&p, want) != hipSuccess) {
      (void)hipGetLastError();
      return PISLAM_ERR_NOMEM;
    }
    cap = want;
    reallocs++;
    if (grew) *grew = true;
    return PISLAM_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    reallocs++;
  }
  template <class T>
  T *as() const { return (T *)p; }
};

}  // namespace

struct pislam_dist_state;   // pislam_dist.inc

struct pislam_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t own_stream = nullptr;    // option "own_stream": a non-blocking stream created and destroyed by the context
  pislam_dist_state *dist = nullptr;   // multi-GPU state (pislam_dist_init), nullptr for a single GPU
  std::string err;
  // staging for host-pointer calls
  DevBuf s_img, s_out, s_pts, s_desc, s_misc, s_rots, s_tmp;
  // compaction scratch (shared by extract and the batch pipeline)
  DevBuf w_cnt, w_off, w_total, w_cellkp;
  // batch pipeline workspace
  DevBuf w_score, w_stage, w_stripcnt, w_work, w_prof, w_ovf, w_stagedesc;
  DevBuf w_ustage, w_ucount;         // bucket selection pass (pf::k_bucket_select): per-unit lists and counts
  // Host-built plan tables the kernels read instead of walking the plan (the bucket selection pass's unit table): one
  // device buffer per distinct CONTENT, never rewritten in place — a captured graph keeps reading the table
  // of the plan it was captured with whatever other shapes the context serves in between.  At most 16 are kept (the oldest
  // is freed, and graphs of the library's pipeline are invalidated through workspace_generation()).
  struct PlanTable {
    std::vector<uint32_t> host;      // (kept: the upload is asynchronous)
    DevBuf dev;
  };
  std::vector<PlanTable *> plan_tables;
  const uint32_t *cur_utab = nullptr;   // the table of the call being issued
  DevBuf w_sync;                     // one-launch path (pf::k_frame): per-pyramid hand-over counters, zero between launches
  // The one-launch path's bounded wait (pf::k_frame): a workgroup that gives up raises a sticky flag in w_sync AND in this
  // host-mapped word, which every call on the context reads first (a plain host load: no synchronisation, no copy).
  uint32_t *frame_flag = nullptr;    // hipHostMalloc'ed, mapped
  uint32_t *frame_flag_dev = nullptr;
  bool frame_disabled = false;       // a timeout was seen: the context takes the three-launch path from then on
  unsigned long long frame_timeouts = 0;   // (counted into workspace_generation: graphs holding a k_frame node are dropped)
  int opt_frame_test = 0;            // test hook: bit 0 = one strip workgroup skips its release, bits 8.. = log2 poll limit
  // pyramid build: all reductions in one launch (pp::k_bilinear_chain) — band counters + [done, sticky fault]; the host-mapped
  // fault word is frame_flag[1]
  DevBuf w_chain;
  size_t chain_words = 0;            // counters the last build laid out (a different layout starts from zeros)
  bool chain_disabled = false;
  int opt_build_chain = 0;           // 1: pislam_pyramid_build_batch runs levels 1.. as ONE launch (pp::k_bilinear_chain); 0 (default): one
                                     // launch per level — measured faster: 150 against 166 us per 64 720p frames (docs/experiments.md)
  int num_cus = 0;
  int opt_pipeline = 0;      // 0 auto, 1 staged (one launch group per level), 2 fused strips
  int opt_dump_score = 0;    // fused pipeline: also materialise the score map (parity hook)
  int opt_strip_rows = 0;    // fused pipeline: strip height override (0 = heuristic)
  int opt_ablate = 0;        // profiling only: skip phases of the fused kernel (results invalid)
  int opt_bucket_round_up = 0;  // profiling: bucket mode always rounds the strip height up to whole bucket rows (round-2 rule)
  int opt_orb_chunks = 0;    // fused pipeline: workgroups per pyramid in k_gather_orb (0 = heuristic)
  uint32_t last_strips = 0;  // strips of the last fused batch call (pislam_frontend_last_stats)
  int opt_repeat_strips = 1; // profiling: launch the strip kernel n times inside the stage-0 event bracket
  int opt_alias = 1;         // fused pipeline: score tile laid over the dead image rows (0 = separate tiles)
  int opt_run_len = 0;       // fused pipeline: strips per workgroup run (0 = default, 1 = independent strips)
  int lanes_in_flight = 1;   // > 1: this context is a lane of a pislam_pipeline of that depth (other batches' kernels share the GPU)
  int opt_lds_pad = 0;       // profiling only: extra dynamic LDS bytes per strip workgroup
  int opt_wgs_per_cu = 0;    // fused pipeline: if > 0, size strip heights for this many workgroups per CU
  int opt_strip_px = 16384;  // profiling: pixels per strip the height heuristic aims at
  int opt_strip_rows_max = 0;    // profiling: upper bound of the heuristic strip height (0 = rule in build_fused_plan)
  int opt_run_order = 1;     // fused pipeline: launch a pyramid's runs longest first (0: in entry order)
  int opt_tile_cols = 0;     // fused pipeline: levels with more classified columns are cut into x-tiles (0 = 704, < 0 = never)
  int opt_bucket_select = 1; // fused pipeline, buckets: 1 = strips as without buckets + pf::k_bucket_select (default), 0 = the strips select (round 1-3)
  int opt_frame = 1;         // fused pipeline: small batches run as ONE launch (pf::k_frame): 1 = batches of 1 or 2 pyramids, n = up to n (<= 8), 0 = never
  int opt_orb_in_strip = 0;  // fused pipeline: 1 = strips describe their own keypoints (measured slower: DESIGN.md §8), 0 = k_gather_orb describes all
  int opt_match_mfma = 1;         // matcher on the matrix cores (0: the VALU popcount kernel)
  int opt_dist_rccl_single = 0;   // test hook: pislam_dist_init(world = 1) still creates a (1-rank) RCCL communicator
  int last_pipeline = 0;
  unsigned last_path = 0;    // PISLAM_PATH_* of the last batch call
  size_t score_bytes_valid = 0;   // bytes of w_score known to be in a consistent (zero-border) state
  pislam_frontend_params last_params{};
  std::vector<pislam_level> last_levels;
  int last_batch = 0;
  size_t last_stride = 0;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  bool timing_valid = false;
  // Sub-batch pipelining of the fused batch path (run_fused): the strip kernels of the sub-batches run back to
  // back on the context stream, the overflow pass + gather/ORB kernel of sub-batch i on `aux_stream` under the
  // strip kernel of sub-batch i+1 (fork / join with events: one call, one context, capturable in a hipGraph).
  static constexpr int MAX_SUB = 16;
  int opt_sub_batches = 1;             // 1 = one launch group (default), n = n sub-batches, 0 = by bytes per sub-batch
  int opt_sub_mb = 0;                  // auto rule: target MiB of pyramids per sub-batch (0 = default)
  hipStream_t aux_stream = nullptr;    // created on first use (non-blocking)
  hipEvent_t ev_sub[MAX_SUB] = {};     // strips of sub-batch i done (context stream -> aux stream)
  hipEvent_t ev_join = nullptr;        // aux stream -> context stream at the end of the call
  int ovf_nsub = 0;                    // layout of w_ovf the last call / reserve established: lists, dwords per list
  size_t ovf_stride = 0;
  unsigned long long ovf_layouts = 0;  // times that layout changed (a replayed graph would read another layout's headers)
  unsigned long long table_uploads = 0;   // plan tables evicted (a graph captured with one of them holds a dangling address)
  // Everything a captured batch call bakes into its kernel arguments besides the caller's own pointers: the
  // addresses of the workspace buffers and the overflow-list layout.  pislam_pipeline_submit replays a hipGraph
  // only while this number is what it was at capture time.
  unsigned long long workspace_generation() const {
    unsigned long long g = ovf_layouts + table_uploads + frame_timeouts;
    for (const DevBuf *b : {&w_cnt, &w_off, &w_total, &w_cellkp, &w_score, &w_stage, &w_stripcnt, &w_work, &w_prof,
                            &w_ovf, &w_stagedesc, &w_ustage, &w_ucount, &w_sync, &w_chain})
      g += b->reallocs;
    return g;
  }
};

namespace {

int fail(pislam_ctx *c, int code, const char *what, hipError_t e = hipSuccess) {
  if (c) {
    c->err = what;
    if (e != hipSuccess) {
      c->err += ": ";
      c->err += hipGetErrorString(e);
    }
  }
  (void)hipGetLastError();
  return code;
}

#define HIPCHK(c, call)                                                   \
  do {                                                                    \
    hipError_t e_ = (call);                                               \
    if (e_ != hipSuccess) return fail((c), PISLAM_ERR_HIP, #call, e_);    \
  } while (0)

#define PCHK(call)                     \
  do {                                 \
    int r_ = (call);                   \
    if (r_ != PISLAM_OK) return r_;    \
  } while (0)

bool is_device_ptr(const void *p) {
  if (!p) return false;
  hipPointerAttribute_t a;
  hipError_t e = hipPointerGetAttributes(&a, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

// Host->device staging of a read-only/in-out byte range.
struct Staged {
  const void *host = nullptr;   // original host pointer (nullptr if the caller's was device)
  void *dev = nullptr;
  size_t bytes = 0;
};

int stage_in(pislam_ctx *c, DevBuf &buf, const void *p, size_t bytes, Staged *s, bool copy = true) {
  s->bytes = bytes;
  if (bytes == 0 || is_device_ptr(p)) {
    s->host = nullptr;
    s->dev = (void *)p;
    return PISLAM_OK;
  }
  if (buf.ensure(bytes) != PISLAM_OK) return fail(c, PISLAM_ERR_NOMEM, "hipMalloc(staging)");
  s->host = p;
  s->dev = buf.p;
  if (copy) HIPCHK(c, hipMemcpyAsync(buf.p, p, bytes, hipMemcpyHostToDevice, c->stream));
  return PISLAM_OK;
}

int stage_out(pislam_ctx *c, const Staged &s, void *host_dst, size_t bytes) {
  if (!s.host || bytes == 0) return PISLAM_OK;
  HIPCHK(c, hipMemcpyAsync(host_dst, s.dev, bytes, hipMemcpyDeviceToHost, c->stream));
  return PISLAM_OK;
}

int sync(pislam_ctx *c) {
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return PISLAM_OK;
}

int launch_ok(pislam_ctx *c, const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(c, PISLAM_ERR_HIP, what, e);
  return PISLAM_OK;
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- the in-grid hand-overs' safety net (pf::k_frame, pp::k_bilinear_chain) --------------------------------------------
// Both kernels contain workgroups that WAIT for other workgroups of the same grid; the waits are bounded, and a workgroup
// that gives up raises a sticky flag in device memory and a word of this host-mapped block (frame_flag[0]: k_frame,
// frame_flag[1]: the build chain), which every call on the context reads first — a plain host load: no synchronisation.
int ensure_fault_flag(pislam_ctx *c) {
  if (c->frame_flag) return PISLAM_OK;
  void *h = nullptr, *d = nullptr;
  if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
    (void)hipGetLastError();
    if (h) (void)hipHostFree(h);
    return fail(c, PISLAM_ERR_NOMEM, "hipHostMalloc(fault flag)");
  }
  c->frame_flag = (uint32_t *)h;
  c->frame_flag_dev = (uint32_t *)d;
  ((volatile uint32_t *)c->frame_flag)[0] = 0;
  ((volatile uint32_t *)c->frame_flag)[1] = 0;
  return PISLAM_OK;
}
// Has a launch of this context given up waiting?  Called at the start of every batch call and pyramid build, by
// pislam_pipeline_submit, pislam_ctx_synchronize, pislam_pipeline_synchronize and pislam_frontend_last_stats.  If so: drain
// the stream, reset the hand-over counters, stop taking that one-launch path on this context (small batches run as three
// launches / the build as one launch per level from now on; captured graphs that hold such a node are dropped through
// workspace_generation) and report the failure ONCE.  The front-end calls that were affected have published
// counts[pyr] = PISLAM_COUNT_INVALID in the caller's own buffer; a build that was affected left levels of its pyramids
// unwritten — the error of THIS call is the notice.
int check_frame_poison(pislam_ctx *c) {
  if (!c->frame_flag) return PISLAM_OK;
  volatile uint32_t *f = (volatile uint32_t *)c->frame_flag;
  const bool frame = f[0] != 0, chain = f[1] != 0;
  if (!frame && !chain) return PISLAM_OK;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->w_sync.p) HIPCHK(c, hipMemsetAsync(c->w_sync.p, 0, c->w_sync.cap, c->stream));
  if (c->w_chain.p) HIPCHK(c, hipMemsetAsync(c->w_chain.p, 0, c->w_chain.cap, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  f[0] = f[1] = 0;
  if (frame) c->frame_disabled = true;
  if (chain) c->chain_disabled = true;
  c->frame_timeouts++;
  if (chain && !frame)
    return fail(c, PISLAM_ERR_HIP, "the one-launch pyramid build (pp::k_bilinear_chain) gave up — a wait for a level's rows timed out, "
                                   "or a frame's workgroups did not share one XCD: the pyramids of that build are invalid; this "
                                   "context now builds one launch per level");
  return fail(c, PISLAM_ERR_HIP, "the one-launch path (pf::k_frame) timed out waiting for its strip workgroups: the affected calls "
                                 "wrote counts = PISLAM_COUNT_INVALID; this context now runs small batches as three launches");
}

// ---- launch helpers shared by the 4-call API and the staged batch path ----

int launch_detect(pislam_ctx *c, const uint8_t *d_img, uint8_t *d_out, int vstep, size_t stride,
                  int batch, int border, int width, int height, int threshold) {
  const int ny = height - 2 * border;
  if (ny <= 0) return PISLAM_OK;   // Fast.h:60: the row loop does not execute
  const int nx = width - 2 * border;
  const int xend = nx > 0 ? border + 16 * cdiv(nx, 16) : border;   // Fast.h:61,149
  dim3 grid(std::max(1, cdiv(xend - border, pk::DT_W)), cdiv(ny, pk::DT_H), batch);
  hipLaunchKernelGGL(pk::k_fast_detect, grid, dim3(256), 0, c->stream, d_img, d_out, vstep, stride,
                     border, width, height, threshold & 0xff, xend);
  return launch_ok(c, "k_fast_detect");
}

int launch_harris(pislam_ctx *c, const uint8_t *d_img, uint8_t *d_out, int vstep, size_t stride,
                  int batch, int border, int width, int height, int32_t threshold) {
  const int ny = height - 2 * border, nx = width - 2 * border;
  if (ny <= 0 || nx <= 0) return PISLAM_OK;
  dim3 grid(cdiv(nx, pk::DT_W), cdiv(ny, pk::DT_H), batch);
  hipLaunchKernelGGL(pk::k_harris_score, grid, dim3(256), 0, c->stream, d_img, d_out, vstep, stride,
                     border, width, height, threshold);
  return launch_ok(c, "k_harris_score");
}

// Ordered extraction of one level for `batch` pyramids.  d_total[b] is the
// running keypoint count of pyramid b (in: offset of this level's first
// keypoint, out: += this level's count).  Scratch: w_cnt / w_off / w_cellkp.
int launch_extract(pislam_ctx *c, const uint8_t *d_score, int vstep, size_t stride, int batch,
                   int border, int lbs, int limit, int width, int height, uint32_t *d_kp,
                   size_t kp_stride, uint32_t cap, uint32_t add_xy, uint32_t *d_total) {
  const int ny = height - 2 * border, nx = width - 2 * border;
  if (ny <= 0 || nx <= 0) return PISLAM_OK;
  if (lbs == 0) {
    const int nrows = cdiv(ny, 2);
    if (c->w_cnt.ensure(sizeof(uint32_t) * (size_t)nrows * batch) != PISLAM_OK ||
        c->w_off.ensure(sizeof(uint32_t) * (size_t)nrows * batch) != PISLAM_OK)
      return fail(c, PISLAM_ERR_NOMEM, "hipMalloc(extract scratch)");
    dim3 grid(cdiv(nrows, 4), 1, batch);
    hipLaunchKernelGGL(pk::k_nms_rows<false>, grid, dim3(256), 0, c->stream, d_score, vstep, stride,
                       border, width, height, nrows, c->w_cnt.as<uint32_t>(), (size_t)nrows,
                       (const uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)0, 0u, 0u);
    PCHK(launch_ok(c, "k_nms_rows<count>"));
    hipLaunchKernelGGL(pk::k_scan_counts, dim3(batch), dim3(256), 0, c->stream,
                       c->w_cnt.as<uint32_t>(), c->w_off.as<uint32_t>(), nrows, (size_t)nrows,
                       d_total);
    PCHK(launch_ok(c, "k_scan_counts"));
    hipLaunchKernelGGL(pk::k_nms_rows<true>, grid, dim3(256), 0, c->stream, d_score, vstep, stride,
                       border, width, height, nrows, (uint32_t *)nullptr, (size_t)nrows,
                       c->w_off.as<uint32_t>(), d_kp, kp_stride, cap, add_xy);
    return launch_ok(c, "k_nms_rows<emit>");
  }
  const int bs = 1 << lbs;
  const int ncx = (nx - 1) / bs + 1;   // Fast.h:201 numBuckets
  const int ncy = (ny - 1) / bs + 1;
  const int ncells = ncx * ncy;
  if (c->w_cnt.ensure(sizeof(uint32_t) * (size_t)ncells * batch) != PISLAM_OK ||
      c->w_off.ensure(sizeof(uint32_t) * (size_t)ncells * batch) != PISLAM_OK ||
      c->w_cellkp.ensure(sizeof(uint32_t) * (size_t)ncells * batch * limit) != PISLAM_OK)
    return fail(c, PISLAM_ERR_NOMEM, "hipMalloc(extract scratch)");
  const size_t lds = sizeof(uint32_t) * (size_t)(bs / 2) * (bs / 2);
  hipLaunchKernelGGL(pk::k_nms_cells, dim3(ncx, ncy, batch), dim3(64), lds, c->stream, d_score, vstep,
                     stride, border, width, height, lbs, limit, ncx, c->w_cnt.as<uint32_t>(),
                     c->w_cellkp.as<uint32_t>(), (size_t)ncells);
  PCHK(launch_ok(c, "k_nms_cells"));
  hipLaunchKernelGGL(pk::k_scan_counts, dim3(batch), dim3(256), 0, c->stream, c->w_cnt.as<uint32_t>(),
                     c->w_off.as<uint32_t>(), ncells, (size_t)ncells, d_total);
  PCHK(launch_ok(c, "k_scan_counts"));
  hipLaunchKernelGGL(pk::k_emit_cells, dim3(cdiv(ncells * limit, 256), 1, batch), dim3(256), 0,
                     c->stream, c->w_cnt.as<uint32_t>(), c->w_off.as<uint32_t>(),
                     c->w_cellkp.as<uint32_t>(), ncells, limit, (size_t)ncells, d_kp, kp_stride, cap,
                     add_xy);
  return launch_ok(c, "k_emit_cells");
}

int check_level_args(pislam_ctx *c, int vstep, int border, int width, int height, int min_border) {
  if (!c) return PISLAM_ERR_INVALID;
  if (vstep <= 0 || width <= 0 || height <= 0 || width > vstep)
    return fail(c, PISLAM_ERR_INVALID, "bad vstep/width/height");
  if (border < min_border) return fail(c, PISLAM_ERR_INVALID, "border too small for this stage");
  return PISLAM_OK;
}

// byte hull [lo, hi) of the image the reference's orbCompute reads for these points
void orb_hull(const uint32_t *pts, size_t n, int vstep, int before, int after_rows, int after_cols, ptrdiff_t *lo,
              ptrdiff_t *hi) {
  ptrdiff_t l = PTRDIFF_MAX, h = PTRDIFF_MIN;
  for (size_t i = 0; i < n; i++) {
    const int x = (pts[i] >> 12) & 0xfff, y = pts[i] & 0xfff;
    const ptrdiff_t a = (ptrdiff_t)(y - before) * vstep + (x - before);
    const ptrdiff_t b = (ptrdiff_t)(y + after_rows) * vstep + (x + after_cols) + 1;
    l = std::min(l, a);
    h = std::max(h, b);
  }
  *lo = l;
  *hi = h;
}

// Stage points + the image hull for the point-list entry points.  On return
// *d_img_base is a device pointer such that d_img_base[y*vstep+x] is valid for
// every byte the kernels touch.
// `before`: rows/columns before a point the consumer reads; `after_rows` / `after_cols`: rows / columns after it
// (ORB: 15 before, rows y+15, columns x+16 incl. the masked column of Orb.h:200-203 — the reference never
// touches row y+16, and with border = 15 that row may lie past the caller's image; Harris 8x8: 3 / 4 / 4).
int stage_points_image(pislam_ctx *c, int vstep, const uint8_t *img, const uint32_t *points, size_t n,
                       const uint8_t **d_img_base, const uint32_t **d_pts, int before = 15,
                       int after_rows = 15, int after_cols = 16) {
  Staged sp;
  std::vector<uint32_t> host_pts;
  const bool pts_dev = is_device_ptr(points);
  const bool img_dev = is_device_ptr(img);
  PCHK(stage_in(c, c->s_pts, points, n * sizeof(uint32_t), &sp));
  *d_pts = (const uint32_t *)sp.dev;
  if (img_dev) {
    *d_img_base = img;
    return PISLAM_OK;
  }
  const uint32_t *hp = points;
  if (pts_dev) {   // need the coordinates on the host to size the hull
    host_pts.resize(n);
    HIPCHK(c, hipMemcpyAsync(host_pts.data(), points, n * sizeof(uint32_t), hipMemcpyDeviceToHost,
                             c->stream));
    PCHK(sync(c));
    hp = host_pts.data();
  }
  ptrdiff_t lo, hi;
  orb_hull(hp, n, vstep, before, after_rows, after_cols, &lo, &hi);
  if (lo < 0) return fail(c, PISLAM_ERR_INVALID, "point too close to the image origin for its patch");
  const size_t bytes = (size_t)(hi - lo);
  if (c->s_img.ensure(bytes) != PISLAM_OK) return fail(c, PISLAM_ERR_NOMEM, "hipMalloc(image hull)");
  HIPCHK(c, hipMemcpyAsync(c->s_img.p, img + lo, bytes, hipMemcpyHostToDevice, c->stream));
  *d_img_base = c->s_img.as<uint8_t>() - lo;
  return PISLAM_OK;
}

}  // namespace

// ===========================================================================
// context
// ===========================================================================
PISLAM_EXPORT int pislam_abi_version(void) { return PISLAM_ABI_VERSION; }

PISLAM_EXPORT int pislam_ctx_create(int device, pislam_ctx **out) {
  if (!out) return PISLAM_ERR_INVALID;
  *out = nullptr;
  // creation failures have no ctx to carry a message: say why on stderr (they are fatal for the
  // caller anyway — there is no CPU fallback)
#define CREATE_CHK(call)                                                                    \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      fprintf(stderr, "pislam_ctx_create: %s -> %s\n", #call, hipGetErrorString(e_));       \
      (void)hipGetLastError();                                                              \
      return PISLAM_ERR_HIP;                                                                \
    }                                                                                       \
  } while (0)
  int ndev = 0;
  CREATE_CHK(hipGetDeviceCount(&ndev));
  if (ndev <= 0) {
    fprintf(stderr, "pislam_ctx_create: no HIP device\n");
    return PISLAM_ERR_HIP;
  }
  if (device < 0) CREATE_CHK(hipGetDevice(&device));
  if (device >= ndev) return PISLAM_ERR_INVALID;
  CREATE_CHK(hipSetDevice(device));
  pislam_ctx *c = new pislam_ctx();
  c->device = device;
  if (hipDeviceGetAttribute(&c->num_cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || c->num_cus <= 0)
    c->num_cus = 256;
  for (auto &e : c->ev) {
    hipError_t r = hipEventCreate(&e);
    if (r != hipSuccess) {
      fprintf(stderr, "pislam_ctx_create: hipEventCreate -> %s\n", hipGetErrorString(r));
      delete c;
      return PISLAM_ERR_HIP;
    }
  }
#undef CREATE_CHK
  *out = c;
  return PISLAM_OK;
}

PISLAM_EXPORT int pislam_dist_finalize(pislam_ctx *c);

PISLAM_EXPORT int pislam_ctx_destroy(pislam_ctx *c) {
  if (!c) return PISLAM_ERR_INVALID;
  (void)hipSetDevice(c->device);
  (void)pislam_dist_finalize(c);
  (void)hipStreamSynchronize(c->stream);
  for (DevBuf *b : {&c->s_img, &c->s_out, &c->s_pts, &c->s_desc, &c->s_misc, &c->s_rots, &c->s_tmp, &c->w_cnt,
                    &c->w_off, &c->w_total, &c->w_cellkp, &c->w_score, &c->w_stage, &c->w_stripcnt, &c->w_work, &c->w_prof, &c->w_ovf,
                    &c->w_stagedesc, &c->w_ustage, &c->w_ucount, &c->w_sync, &c->w_chain})
    b->release();
  for (auto *t : c->plan_tables) {
    t->dev.release();
    delete t;
  }
  c->plan_tables.clear();
  for (auto &e : c->ev)
    if (e) (void)hipEventDestroy(e);
  if (c->aux_stream) {
    (void)hipStreamSynchronize(c->aux_stream);
    (void)hipStreamDestroy(c->aux_stream);
  }
  for (auto &e : c->ev_sub)
    if (e) (void)hipEventDestroy(e);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  if (c->frame_flag) (void)hipHostFree(c->frame_flag);
  delete c;
  return PISLAM_OK;
}

PISLAM_EXPORT int pislam_ctx_set_stream(pislam_ctx *c, void *s) {
  if (!c) return PISLAM_ERR_INVALID;
  c->stream = (hipStream_t)s;
  return PISLAM_OK;
}

namespace {
// mode 1: a stream with default flags — it still synchronises with the legacy null stream, so device-pointer
// inputs produced on the null stream stay ordered before the call, as they were when the context issued on
// the null stream itself; mode 2: hipStreamNonBlocking (the caller orders its producers explicitly).
int use_own_stream(pislam_ctx *c, int mode) {
  HIPCHK(c, hipSetDevice(c->device));
  if (mode) {
    if (c->own_stream) {
      HIPCHK(c, hipStreamSynchronize(c->own_stream));
      if (c->stream == c->own_stream) c->stream = nullptr;
      HIPCHK(c, hipStreamDestroy(c->own_stream));
      c->own_stream = nullptr;
    }
    HIPCHK(c, hipStreamCreateWithFlags(&c->own_stream, mode == 2 ? hipStreamNonBlocking : hipStreamDefault));
    c->stream = c->own_stream;
  } else if (c->own_stream) {
    HIPCHK(c, hipStreamSynchronize(c->own_stream));
    if (c->stream == c->own_stream) c->stream = nullptr;
    HIPCHK(c, hipStreamDestroy(c->own_stream));
    c->own_stream = nullptr;
  }
  return PISLAM_OK;
}
}  // namespace

PISLAM_EXPORT int pislam_ctx_set_option(pislam_ctx *c, const char *key, int value) {
  if (!c || !key) return PISLAM_ERR_INVALID;
  if (!strcmp(key, "pipeline")) {
    if (value < 0 || value > 2) return fail(c, PISLAM_ERR_INVALID, "pipeline must be 0 (auto), 1 (staged) or 2 (fused)");
    c->opt_pipeline = value;
  } else if (!strcmp(key, "own_stream")) {
    if (value < 0 || value > 2) return fail(c, PISLAM_ERR_INVALID, "own_stream must be 0, 1 (default flags) or 2 (non-blocking)");
    return use_own_stream(c, value);
  } else if (!strcmp(key, "dump_score")) {
    c->opt_dump_score = value != 0;
  } else if (!strcmp(key, "repeat_strips")) {
    if (value < 1 || value > 64) return fail(c, PISLAM_ERR_INVALID, "repeat_strips must be 1..64");
    c->opt_repeat_strips = value;
  } else if (!strcmp(key, "alias")) {
    c->opt_alias = value != 0;
  } else if (!strcmp(key, "run_len")) {
    if (value < 0 || value > 64) return fail(c, PISLAM_ERR_INVALID, "run_len must be 0 (default) .. 64");
    c->opt_run_len = value;
  } else if (!strcmp(key, "lds_pad")) {
    c->opt_lds_pad = value < 0 ? 0 : value;
  } else if (!strcmp(key, "wgs_per_cu")) {
    if (value < 0 || value > 8) return fail(c, PISLAM_ERR_INVALID, "wgs_per_cu must be 0..8");
    c->opt_wgs_per_cu = value;
  } else if (!strcmp(key, "match_mfma")) {
    c->opt_match_mfma = value != 0;
  } else if (!strcmp(key, "run_order")) {
    c->opt_run_order = value != 0;
  } else if (!strcmp(key, "strip_px")) {
    c->opt_strip_px = std::max(4096, value);
  } else if (!strcmp(key, "strip_rows_max")) {
    c->opt_strip_rows_max = value <= 0 ? 0 : std::max(16, std::min(64, value & ~1));
  } else if (!strcmp(key, "tile_cols")) {
    if (value > 0 && value < 64) return fail(c, PISLAM_ERR_INVALID, "tile_cols must be 0 (default), < 0 (never) or >= 64");
    c->opt_tile_cols = value;
  } else if (!strcmp(key, "frame")) {   // 0: never one launch; 1 (default): batches of 1 or 2 pyramids; n = 2..8: batches of up to n
    if (value < 0 || value > 8) return fail(c, PISLAM_ERR_INVALID, "frame must be 0..8");
    c->opt_frame = value;
  } else if (!strcmp(key, "frame_test")) {   // test hook of the one-launch path's bounded wait (see pf::k_frame `test`)
    c->opt_frame_test = value;
  } else if (!strcmp(key, "frame_rearm")) {  // 1: take the one-launch paths again after a reported timeout (tests)
    if (value) c->frame_disabled = c->chain_disabled = false;
  } else if (!strcmp(key, "build_chain")) {  // pyramid build: 1 = levels 1.. in ONE launch (pp::k_bilinear_chain), 0 (default) one launch per level
    c->opt_build_chain = value != 0;
  } else if (!strcmp(key, "orb_in_strip")) {
    c->opt_orb_in_strip = value != 0;
  } else if (!strcmp(key, "bucket_select")) {
    c->opt_bucket_select = value != 0;
  } else if (!strcmp(key, "dist_rccl_single")) {
    c->opt_dist_rccl_single = value != 0;
  } else if (!strcmp(key, "bucket_round_up")) {
    c->opt_bucket_round_up = value != 0;
  } else if (!strcmp(key, "orb_chunks")) {
    if (value < 0 || value > 1024) return fail(c, PISLAM_ERR_INVALID, "orb_chunks must be 0..1024");
    c->opt_orb_chunks = value;
  } else if (!strcmp(key, "sub_batches")) {
    if (value < 0 || value > pislam_ctx::MAX_SUB) return fail(c, PISLAM_ERR_INVALID, "sub_batches must be 0 (auto) .. 16");
    c->opt_sub_batches = value;
  } else if (!strcmp(key, "sub_mb")) {
    c->opt_sub_mb = std::max(0, value);
  } else if (!strcmp(key, "ablate")) {
    c->opt_ablate = value;
  } else if (!strcmp(key, "strip_rows")) {
    if (value < 0 || value > 64 || (value & 1)) return fail(c, PISLAM_ERR_INVALID, "strip_rows must be even, 0..64");
    c->opt_strip_rows = value;
  } else {
    return fail(c, PISLAM_ERR_INVALID, "unknown option");
  }
  return PISLAM_OK;
}

PISLAM_EXPORT int pislam_ctx_synchronize(pislam_ctx *c) {
  if (!c) return PISLAM_ERR_INVALID;
  PCHK(sync(c));
  return check_frame_poison(c);
}

PISLAM_EXPORT const char *pislam_last_error(const pislam_ctx *c) { return c ? c->err.c_str() : "null ctx"; }

PISLAM_EXPORT const int8_t *pislam_brief_table(void) {
  static int8_t tab[30 * 256 * 4];
  static bool ready = false;
  if (!ready) {
    for (int i = 0; i < 30 * 256; i++)
      for (int k = 0; k < 4; k++) tab[i * 4 + k] = (int8_t)((h_brief_tab_packed[i] >> (8 * k)) & 0xff);
    ready = true;
  }
  return tab;
}

PISLAM_EXPORT size_t pislam_centroids_size(size_t n) { return (2 * n + 7) & ~(size_t)7; }

// ===========================================================================
// the four reference entry points
// ===========================================================================
PISLAM_EXPORT int pislam_fast_detect(pislam_ctx *c, int vstep, int border, int width, int height,
                                     const uint8_t *img, uint8_t *out, int threshold) {
  PCHK(check_level_args(c, vstep, border, width, height, 3));
  if (!img || !out) return fail(c, PISLAM_ERR_INVALID, "null image");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t bytes = (size_t)height * vstep;
  // The over-classified columns [width-border, xend) may run past vstep, i.e. into the next row (flat
  // addressing, as the reference's 16-byte vectors do): with a small border the ring of the last
  // classified row then reaches a few bytes beyond height*vstep — bytes the reference reads too.
  size_t img_bytes = bytes;
  if (height > 2 * border && width > 2 * border) {
    const size_t xend = (size_t)border + 16 * (size_t)cdiv(width - 2 * border, 16);
    img_bytes = std::max(bytes, (size_t)(height - border + 2) * vstep + xend + 3);
  }
  Staged si, so;
  PCHK(stage_in(c, c->s_img, img, img_bytes, &si));
  PCHK(stage_in(c, c->s_out, out, bytes, &so));
  PCHK(launch_detect(c, (const uint8_t *)si.dev, (uint8_t *)so.dev, vstep, 0, 1, border, width, height,
                     threshold));
  PCHK(stage_out(c, so, out, bytes));
  if (si.host || so.host) PCHK(sync(c));
  return PISLAM_OK;
}

PISLAM_EXPORT int pislam_fast_score_harris(pislam_ctx *c, int vstep, int border, int width, int height,
                                           const uint8_t *img, int32_t threshold, uint8_t *out) {
  PCHK(check_level_args(c, vstep, border, width, height, 4));
  if (!img || !out) return fail(c, PISLAM_ERR_INVALID, "null image");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t bytes = (size_t)height * vstep;
  Staged si, so;
  PCHK(stage_in(c, c->s_img, img, bytes, &si));
  PCHK(stage_in(c, c->s_out, out, bytes, &so));
  PCHK(launch_harris(c, (const uint8_t *)si.dev, (uint8_t *)so.dev, vstep, 0, 1, border, width, height,
                     threshold));
  PCHK(stage_out(c, so, out, bytes));
  if (si.host || so.host) PCHK(sync(c));
  return PISLAM_OK;
}

PISLAM_EXPORT int pislam_fast_extract(pislam_ctx *c, int vstep, int border, int lbs, int limit, int width,
                                      int height, const uint8_t *out, uint32_t *results, size_t capacity,
                                      size_t *count) {
  PCHK(check_level_args(c, vstep, border, width, height, 1));
  if (!out || !count || (!results && capacity)) return fail(c, PISLAM_ERR_INVALID, "null pointer");
  if (lbs < 0 || lbs > 8 || limit < 1 || limit > 64)
    return fail(c, PISLAM_ERR_INVALID, "logBucketSize must be 0..8 and bucketLimit 1..64");
  if (width > 4096 || height > 4096)
    return fail(c, PISLAM_ERR_INVALID, "encodeFast holds 12-bit coordinates (Util.h:27-29)");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t bytes = (size_t)height * vstep;
  Staged so, sr;
  PCHK(stage_in(c, c->s_out, out, bytes, &so));
  PCHK(stage_in(c, c->s_pts, results, capacity * sizeof(uint32_t), &sr, /*copy=*/false));
  if (c->w_total.ensure(sizeof(uint32_t)) != PISLAM_OK) return fail(c, PISLAM_ERR_NOMEM, "hipMalloc");
  HIPCHK(c, hipMemsetAsync(c->w_total.p, 0, sizeof(uint32_t), c->stream));
  const uint32_t cap32 = (uint32_t)std::min<size_t>(capacity, 0xffffffffu);
  PCHK(launch_extract(c, (const uint8_t *)so.dev, vstep, 0, 1, border, lbs, limit, width, height,
                      (uint32_t *)sr.dev, 0, cap32, 0u, c->w_total.as<uint32_t>()));
  uint32_t n = 0;
  HIPCHK(c, hipMemcpyAsync(&n, c->w_total.p, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  PCHK(sync(c));
  *count = n;
  PCHK(stage_out(c, sr, results, std::min<size_t>(n, capacity) * sizeof(uint32_t)));
  if (sr.host) PCHK(sync(c));
  return PISLAM_OK;
}

namespace {
int orb_common(pislam_ctx *c, int mode, int vstep, int words, const uint8_t *img, const uint32_t *points,
               const uint8_t *rots, size_t n, uint32_t *descriptors, int32_t *centroids) {
  if (!c) return PISLAM_ERR_INVALID;
  if (vstep <= 0) return fail(c, PISLAM_ERR_INVALID, "bad vstep");
  if (mode != 1 && (words < 1 || words > 8)) return fail(c, PISLAM_ERR_INVALID, "words must be 1..8");
  HIPCHK(c, hipSetDevice(c->device));
  if (mode == 1) {
    const size_t n8 = pislam_centroids_size(n);
    Staged sc;
    PCHK(stage_in(c, c->s_misc, centroids, n8 * sizeof(int32_t), &sc, false));
    if (n8) HIPCHK(c, hipMemsetAsync(sc.dev, 0, n8 * sizeof(int32_t), c->stream));
    if (n) {
      if (!img || !points) return fail(c, PISLAM_ERR_INVALID, "null pointer");
      const uint8_t *d_img;
      const uint32_t *d_pts;
      PCHK(stage_points_image(c, vstep, img, points, n, &d_img, &d_pts));
      hipLaunchKernelGGL(pk::k_orb<1>, dim3(cdiv((int)n, 4)), dim3(256), 0, c->stream, d_img, vstep,
                         (size_t)0, d_pts, (size_t)0, (const uint32_t *)nullptr, (uint32_t)n,
                         (uint32_t)n, 0, (uint32_t *)nullptr, (size_t)0, (int32_t *)sc.dev,
                         (const uint8_t *)nullptr);
      PCHK(launch_ok(c, "k_orb<centroids>"));
    }
    PCHK(stage_out(c, sc, centroids, n8 * sizeof(int32_t)));
    return sync(c);
  }
  if (n == 0) return PISLAM_OK;
  if (!img || !points || !descriptors) return fail(c, PISLAM_ERR_INVALID, "null pointer");
  const uint8_t *d_img;
  const uint32_t *d_pts;
  PCHK(stage_points_image(c, vstep, img, points, n, &d_img, &d_pts));
  Staged sd, sr;
  PCHK(stage_in(c, c->s_desc, descriptors, n * words * sizeof(uint32_t), &sd, false));
  const uint8_t *d_rots = nullptr;
  if (mode == 2) {
    PCHK(stage_in(c, c->s_rots, rots, n, &sr));
    d_rots = (const uint8_t *)sr.dev;
    // rotations outside 0..29 leave the descriptor untouched (Brief.h switch): preserve caller bytes
    if (sd.host)
      HIPCHK(c, hipMemcpyAsync(sd.dev, descriptors, n * words * sizeof(uint32_t), hipMemcpyHostToDevice,
                               c->stream));
    hipLaunchKernelGGL(pk::k_orb<2>, dim3(cdiv((int)n, 4)), dim3(256), 0, c->stream, d_img, vstep,
                       (size_t)0, d_pts, (size_t)0, (const uint32_t *)nullptr, (uint32_t)n, (uint32_t)n,
                       words, (uint32_t *)sd.dev, (size_t)0, (int32_t *)nullptr, d_rots);
  } else {
    hipLaunchKernelGGL(pk::k_orb<0>, dim3(cdiv((int)n, 4)), dim3(256), 0, c->stream, d_img, vstep,
                       (size_t)0, d_pts, (size_t)0, (const uint32_t *)nullptr, (uint32_t)n, (uint32_t)n,
                       words, (uint32_t *)sd.dev, (size_t)0, (int32_t *)nullptr,
                       (const uint8_t *)nullptr);
  }
  PCHK(launch_ok(c, "k_orb"));
  PCHK(stage_out(c, sd, descriptors, n * words * sizeof(uint32_t)));
  return sync(c);
}
}  // namespace

PISLAM_EXPORT int pislam_orb_compute(pislam_ctx *c, int vstep, int words, const uint8_t *img,
                                     const uint32_t *points, size_t n, uint32_t *descriptors) {
  return orb_common(c, 0, vstep, words, img, points, nullptr, n, descriptors, nullptr);
}

PISLAM_EXPORT int pislam_orb_centroids(pislam_ctx *c, int vstep, const uint8_t *img, const uint32_t *points,
                                       size_t n, int32_t *centroids) {
  if (!c) return PISLAM_ERR_INVALID;
  if (!centroids && n) return fail(c, PISLAM_ERR_INVALID, "null pointer");
  return orb_common(c, 1, vstep, 0, img, points, nullptr, n, nullptr, centroids);
}

PISLAM_EXPORT int pislam_brief_describe(pislam_ctx *c, int vstep, int words, const uint8_t *img,
                                        const uint32_t *points, const uint8_t *rots, size_t n,
                                        uint32_t *descriptors) {
  if (!c) return PISLAM_ERR_INVALID;
  if (!rots && n) return fail(c, PISLAM_ERR_INVALID, "null pointer");
  return orb_common(c, 2, vstep, words, img, points, rots, n, descriptors, nullptr);
}

PISLAM_EXPORT int pislam_orb_angles(pislam_ctx *c, const int32_t *centroids, size_t n8, uint8_t *angles) {
  if (!c) return PISLAM_ERR_INVALID;
  if (n8 % 8) return fail(c, PISLAM_ERR_INVALID, "n8 must be a multiple of 8 (Orb.h:314)");
  if (n8 == 0) return PISLAM_OK;
  if (!centroids || !angles) return fail(c, PISLAM_ERR_INVALID, "null pointer");
  HIPCHK(c, hipSetDevice(c->device));
  Staged sc, sa;
  PCHK(stage_in(c, c->s_misc, centroids, n8 * sizeof(int32_t), &sc));
  PCHK(stage_in(c, c->s_rots, angles, n8 / 2, &sa, false));
  hipLaunchKernelGGL(pk::k_angles, dim3(cdiv((int)(n8 / 2), 256)), dim3(256), 0, c->stream,
                     (const int32_t *)sc.dev, (int)(n8 / 2), (uint8_t *)sa.dev);
  PCHK(launch_ok(c, "k_angles"));
  PCHK(stage_out(c, sa, angles, n8 / 2));
  return sync(c);
}

PISLAM_EXPORT int pislam_harris_score_points(pislam_ctx *c, int vstep, const uint8_t *img,
                                             const uint32_t *points, size_t n, int32_t threshold,
                                             uint8_t *scores) {
  if (!c) return PISLAM_ERR_INVALID;
  if (n == 0) return PISLAM_OK;
  if (!img || !points || !scores || vstep <= 0) return fail(c, PISLAM_ERR_INVALID, "bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  const uint8_t *d_img;
  const uint32_t *d_pts;
  PCHK(stage_points_image(c, vstep, img, points, n, &d_img, &d_pts, 3, 4, 4));   // Harris.h:102-110
  Staged ss;
  PCHK(stage_in(c, c->s_rots, scores, n, &ss, false));
  hipLaunchKernelGGL(pk::k_harris_points, dim3(cdiv((int)n, 256)), dim3(256), 0, c->stream, d_img, vstep,
                     d_pts, (int)n, threshold, (uint8_t *)ss.dev);
  PCHK(launch_ok(c, "k_harris_points"));
  PCHK(stage_out(c, ss, scores, n));
  return sync(c);
}



namespace {
// launch bilinear N/M for `batch` images; fast 4-block kernel when everything is 16-byte aligned
template <int N, int M>
int launch_bilinear(pislam_ctx *c, const uint8_t *src, uint8_t *dst, int vstep_src, int vstep_dst, size_t stride_src,
                    size_t stride_dst, int batch, int width, int height) {
  const int nbx = cdiv(width, N), nby = cdiv(height, N);
  const bool fast = ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 4 == 0) && vstep_src % 16 == 0 && vstep_dst % 4 == 0 &&
                    stride_src % 16 == 0 && stride_dst % 4 == 0 &&
                    (ptrdiff_t)cdiv(nbx, 4) * 4 * N <= vstep_src;      // the last group's 16-byte loads stay in the row
  if (fast)
    hipLaunchKernelGGL((pp::k_bilinear4<N, M>), dim3(cdiv(cdiv(nbx, 4) * nby * M, 256), 1, batch), dim3(256), 0,
                       c->stream, src, dst, vstep_src, vstep_dst, stride_src, stride_dst, width, height);
  else
    hipLaunchKernelGGL((pp::k_bilinear<N, M>), dim3(cdiv(nbx * M, 256), cdiv(nby * M, 4), batch), dim3(256), 0, c->stream,
                       src, dst, vstep_src, vstep_dst, stride_src, stride_dst, width, height);
  return launch_ok(c, "k_bilinear");
}
}  // namespace

// ===========================================================================
// image preparation ("next" tier): gaussian5x5, bilinear7_8, bilinear13_16
// ===========================================================================
namespace {
// kind 0 = gaussian5x5, 1 = bilinear7_8, 2 = bilinear13_16
int prep_common(pislam_ctx *c, int kind, int vstep, int width, int height, const uint8_t *img, uint8_t *out) {
  if (!c) return PISLAM_ERR_INVALID;
  if (!img || !out) return fail(c, PISLAM_ERR_INVALID, "null image");
  if (vstep <= 0 || width <= 0 || height <= 0) return fail(c, PISLAM_ERR_INVALID, "bad vstep/width/height");
  const int N = kind == 0 ? 1 : (kind == 1 ? 8 : 16);
  const int wpad = (width + N - 1) / N * N, hpad = (height + N - 1) / N * N;   // Bilinear.h:32,155 padding
  if (wpad > vstep) return fail(c, PISLAM_ERR_INVALID, "width (padded to the block size) exceeds vstep");
  if (kind == 0 && (width < 3 || height < 3)) return fail(c, PISLAM_ERR_INVALID, "gaussian5x5 needs at least 3x3");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t in_bytes = (size_t)hpad * vstep;
  const int M = kind == 1 ? 7 : 13;
  const size_t out_rows = kind == 0 ? (size_t)height : (size_t)(hpad / N) * M;
  const size_t out_bytes = out_rows * vstep;
  Staged si, so;
  PCHK(stage_in(c, c->s_img, img, in_bytes, &si));
  const bool inplace = img == out;
  const uint8_t *d_src = (const uint8_t *)si.dev;
  uint8_t *d_dst;
  if (inplace) {
    // results are a pure function of the ORIGINAL image (as the reference's in-place loops are):
    // run from a private copy of the source
    if (c->s_tmp.ensure(in_bytes) != PISLAM_OK) return fail(c, PISLAM_ERR_NOMEM, "hipMalloc(prep temp)");
    HIPCHK(c, hipMemcpyAsync(c->s_tmp.p, d_src, in_bytes, hipMemcpyDeviceToDevice, c->stream));
    d_src = c->s_tmp.as<uint8_t>();
    d_dst = (uint8_t *)si.dev;
    so = si;
  } else {
    PCHK(stage_in(c, c->s_out, out, std::max(out_bytes, (size_t)1), &so));   // keep untouched bytes
    d_dst = (uint8_t *)so.dev;
  }
  if (kind == 0) {
    dim3 grid(cdiv(width, pp::G_TW), cdiv(height, pp::G_TH), 1);
    hipLaunchKernelGGL(pp::k_gaussian5x5, grid, dim3(256), 0, c->stream, d_src, d_dst, vstep, vstep, (size_t)0,
                       (size_t)0, width, height);
    PCHK(launch_ok(c, "k_gaussian5x5"));
  } else if (kind == 1) {
    PCHK((launch_bilinear<8, 7>(c, d_src, d_dst, vstep, vstep, 0, 0, 1, width, height)));
  } else {
    PCHK((launch_bilinear<16, 13>(c, d_src, d_dst, vstep, vstep, 0, 0, 1, width, height)));
  }
  if (so.host) {
    PCHK(stage_out(c, so, out, out_bytes));   // rows the kernels can have written
    PCHK(sync(c));
  } else if (si.host) {
    PCHK(sync(c));
  }
  return PISLAM_OK;
}
}  // namespace

PISLAM_EXPORT int pislam_gaussian5x5(pislam_ctx *c, int vstep, int width, int height, const uint8_t *img,
                                     uint8_t *out) {
  return prep_common(c, 0, vstep, width, height, img, out);
}
PISLAM_EXPORT int pislam_bilinear7_8(pislam_ctx *c, int vstep, int width, int height, const uint8_t *img,
                                     uint8_t *out) {
  return prep_common(c, 1, vstep, width, height, img, out);
}
PISLAM_EXPORT int pislam_bilinear13_16(pislam_ctx *c, int vstep, int width, int height, const uint8_t *img,
                                       uint8_t *out) {
  return prep_common(c, 2, vstep, width, height, img, out);
}


// ---------------------------------------------------------------------------
// on-GPU pyramid build (BASELINE config 5): frame -> gaussian5x5 -> level 0, then a chain of
// bilinear7_8 / bilinear13_16 reductions, every level written into a vertically stacked pyramid.
// ---------------------------------------------------------------------------
PISLAM_EXPORT int pislam_pyramid_layout(int width, int height, int nlevels, const int32_t *steps, int vstep_min,
                                        pislam_level *levels, int32_t *vstep, int32_t *rows) {
  if (width <= 0 || height <= 0 || nlevels < 1 || nlevels > 16 || !levels || (nlevels > 1 && !steps))
    return PISLAM_ERR_INVALID;
  int w = width, h = height, row = 0, maxcols = width;
  for (int l = 0; l < nlevels; l++) {
    int written = 0;                       // rows the reduction INTO this level writes (whole 7x7 / 13x13 blocks)
    if (l > 0) {
      if (steps[l - 1] == 1) {
        written = (h + 7) / 8 * 7;
        maxcols = std::max(maxcols, (w + 7) / 8 * 7);
        w = w * 7 / 8;                     // Bilinear.h:34-35: round down
        h = h * 7 / 8;
      } else if (steps[l - 1] == 2) {
        written = (h + 15) / 16 * 13;
        maxcols = std::max(maxcols, (w + 15) / 16 * 13);
        w = w * 13 / 16;                   // Bilinear.h:157-158
        h = h * 13 / 16;
      } else {
        return PISLAM_ERR_INVALID;
      }
    }
    if (w < 3 || h < 3) return PISLAM_ERR_INVALID;
    levels[l].width = w;
    levels[l].height = h;
    levels[l].row0 = row;
    levels[l].col0 = 0;
    // the slot holds the padding rows the next reduction reads (Bilinear.h:32,155) and every row the
    // reduction into this level writes
    row += std::max((h + 15) / 16 * 16, written);
  }
  if (vstep) *vstep = std::max(vstep_min, (maxcols + 15) / 16 * 16);
  if (rows) *rows = row;
  return PISLAM_OK;
}

PISLAM_EXPORT int pislam_pyramid_build_batch(pislam_ctx *c, int nlevels, const int32_t *steps,
                                             const pislam_level *levels, const uint8_t *frames, int frame_vstep,
                                             size_t frame_stride, int batch, uint8_t *pyramids, int vstep,
                                             int rows, size_t pyramid_stride, int flags) {
  const bool blur = (flags & PISLAM_BUILD_BLUR) != 0;
  if (!c) return PISLAM_ERR_INVALID;
  // (ABI 1 took `blur` = any non-zero value here: unknown bits are refused, not silently read as flags)
  if (flags & ~(PISLAM_BUILD_BLUR | PISLAM_BUILD_MARGINS_CLEAN | PISLAM_BUILD_CHECK_MARGINS))
    return fail(c, PISLAM_ERR_INVALID, "unknown PISLAM_BUILD_* flag bits");
  if (!levels || !frames || !pyramids || batch <= 0 || nlevels < 1 || nlevels > 16 || (nlevels > 1 && !steps))
    return fail(c, PISLAM_ERR_INVALID, "bad argument");
  if (!is_device_ptr(frames) || !is_device_ptr(pyramids))
    return fail(c, PISLAM_ERR_INVALID, "the pyramid builder takes device pointers only");
  PCHK(check_frame_poison(c));
  for (int l = 0; l < nlevels; l++) {
    const int pad = l + 1 < nlevels ? (steps[l] == 1 ? 8 : 16) : 1;
    const int wp = (levels[l].width + pad - 1) / pad * pad, hp = (levels[l].height + pad - 1) / pad * pad;
    if (levels[l].col0 != 0 || wp > vstep || levels[l].row0 + hp > rows || pyramid_stride < (size_t)rows * vstep)
      return fail(c, PISLAM_ERR_INVALID, "level (with its bilinear padding) does not fit the pyramid buffer");
  }
  for (int l = 0; l + 1 < nlevels; l++) {           // whole output blocks must land inside the next level's slot
    const int N = steps[l] == 1 ? 8 : 16, M = steps[l] == 1 ? 7 : 13;
    const int oh = (levels[l].height + N - 1) / N * M, ow = (levels[l].width + N - 1) / N * M;
    const int slot_end = l + 2 < nlevels ? levels[l + 2].row0 : rows;
    if (ow > vstep || levels[l + 1].row0 + oh > slot_end)
      return fail(c, PISLAM_ERR_INVALID, "a reduction's output blocks overrun the next level's slot (use pislam_pyramid_layout)");
  }
  if (levels[0].width > frame_vstep || frame_stride < (size_t)levels[0].height * frame_vstep)
    return fail(c, PISLAM_ERR_INVALID, "frame buffer too small");
  HIPCHK(c, hipSetDevice(c->device));
  // Padding bytes are read by the bilinear steps (block padding) and by FAST's right-edge columns: they are
  // defined as zero.  Every build rewrites the same rectangle of each level's slot and zeroes the margins
  // around it that those consumers read (pp::k_zero_margins) — on EVERY call: a caller may have scribbled over
  // the buffer, or the allocator may hand out a recycled address — unless the caller vouches for them
  // (PISLAM_BUILD_MARGINS_CLEAN: a buffer this function filled before with the same layout and nobody wrote
  // to since; the margin pass costs ~20 us per 64 720p frames).  Bytes beyond the margins are nobody's input
  // and are left untouched (zero-initialise the buffer once if they must be defined).
  const bool clean = (flags & PISLAM_BUILD_MARGINS_CLEAN) != 0, check = (flags & PISLAM_BUILD_CHECK_MARGINS) != 0;
  if (!clean || check) {
    pp::ZeroPlan Z;
    memset(&Z, 0, sizeof(Z));
    Z.nlevels = nlevels;
    Z.vstep = vstep;
    for (int l = 0; l < nlevels; l++) {
      Z.row0[l] = levels[l].row0;
      Z.slot_rows[l] = (l + 1 < nlevels ? levels[l + 1].row0 : rows) - levels[l].row0;
      if (l == 0) {
        Z.ww[l] = levels[0].width;
        Z.wh[l] = levels[0].height;
      } else {
        const int N = steps[l - 1] == 1 ? 8 : 16, M = steps[l - 1] == 1 ? 7 : 13;
        Z.ww[l] = (levels[l - 1].width + N - 1) / N * M;
        Z.wh[l] = (levels[l - 1].height + N - 1) / N * M;
      }
    }
    if (clean) {
      // debug: verify the caller's promise instead of trusting it (synchronises; a dirty margin is an error)
      if (c->w_total.ensure(sizeof(uint32_t)) != PISLAM_OK) return fail(c, PISLAM_ERR_NOMEM, "hipMalloc");
      HIPCHK(c, hipMemsetAsync(c->w_total.p, 0, sizeof(uint32_t), c->stream));
      hipLaunchKernelGGL(pp::k_zero_margins<true>, dim3(2 * nlevels, batch), dim3(256), 0, c->stream, Z, pyramids,
                         pyramid_stride, c->w_total.as<unsigned int>());
      PCHK(launch_ok(c, "k_zero_margins<check>"));
      uint32_t nz = 0;
      HIPCHK(c, hipMemcpyAsync(&nz, c->w_total.p, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
      PCHK(sync(c));
      if (nz) return fail(c, PISLAM_ERR_INVALID, "PISLAM_BUILD_MARGINS_CLEAN was passed but the margins hold non-zero bytes");
    } else {
      hipLaunchKernelGGL(pp::k_zero_margins<false>, dim3(2 * nlevels, batch), dim3(256), 0, c->stream, Z, pyramids,
                         pyramid_stride, (unsigned int *)nullptr);
      PCHK(launch_ok(c, "k_zero_margins"));
    }
  }
  const int w0 = levels[0].width, h0 = levels[0].height;
  if (blur) {
    hipLaunchKernelGGL(pp::k_gaussian5x5, dim3(cdiv(w0, pp::G_TW), cdiv(h0, pp::G_TH), batch), dim3(256), 0, c->stream,
                       frames, pyramids + (size_t)levels[0].row0 * vstep, frame_vstep, vstep, frame_stride,
                       pyramid_stride, w0, h0);
    PCHK(launch_ok(c, "k_gaussian5x5"));
  } else {
    HIPCHK(c, hipMemcpy2DAsync(pyramids + (size_t)levels[0].row0 * vstep, vstep, frames, frame_vstep, w0, h0,
                               hipMemcpyDeviceToDevice, c->stream));
    for (int b = 1; b < batch; b++)
      HIPCHK(c, hipMemcpy2DAsync(pyramids + b * pyramid_stride + (size_t)levels[0].row0 * vstep, vstep,
                                 frames + b * frame_stride, frame_vstep, w0, h0, hipMemcpyDeviceToDevice, c->stream));
  }
  // (Fusing two chained reductions per launch — a 128-tile of level k maps onto whole blocks of 13/16 then 7/8,
  //  level k+1 kept in LDS — was built and measured: 184 vs 167 us per 64 frames, 667 vs 584 us per 256.  The
  //  64-frame pyramid set fits the 256 MiB Infinity Cache, so the re-read the fusion saves never reaches HBM,
  //  and the LDS round trips cost more than k_bilinear4's register-only path.)
  // (The small levels in ONE launch — one 1024-thread workgroup per pyramid walking levels 3 .. 7 through k_bilinear4's
  //  work items, a device-scope fence + barrier between levels — was built in round 4, bit-exact, and measured: the step of
  //  64 720p frames 0.375 -> 0.53 ms from level 3 on, 0.59 ms from level 2 on: 64 workgroups with a handful of dependent
  //  load -> compute -> store round trips each are far slower than five launches that fill the chip, launch floors included.)
  // Round 6: ALL reductions as one launch with a row-band hand-over between the levels (pp::k_bilinear_chain) — the same
  // work items as the per-level launches, every workgroup starting as soon as the source rows it reads are complete.
  bool chain = c->opt_build_chain && !c->chain_disabled && nlevels >= 3;
  pp::ChainPlan C;
  memset(&C, 0, sizeof(C));
  if (chain) {
    chain = (uintptr_t)pyramids % 16 == 0 && vstep % 16 == 0 && pyramid_stride % 16 == 0;   // k_bilinear4's fast path, every level
    C.nlevels = nlevels;
    C.vstep = vstep;
    C.batch = batch;
    C.groups = cdiv(batch, 8);
    long wg = 0;
    int bands = 0;
    for (int l = 1; l < nlevels && chain; l++) {
      const int N = steps[l - 1] == 1 ? 8 : 16, M = steps[l - 1] == 1 ? 7 : 13;
      const int sw = levels[l - 1].width, sh = levels[l - 1].height;
      const int nbx = cdiv(sw, N), nby = cdiv(sh, N);
      if ((ptrdiff_t)cdiv(nbx, 4) * 4 * N > vstep) chain = false;      // the last group's 16-byte loads must stay in the row
      C.kind[l] = steps[l - 1];
      C.row0[l] = levels[l].row0;
      C.sw[l] = sw;
      C.sh[l] = sh;
      C.nq[l] = cdiv(nbx, 4);
      C.oh[l] = nby * M;
      C.wpf[l] = cdiv(C.nq[l] * C.oh[l], 256);
      C.wg0[l] = (int)wg;
      wg += 8L * C.wpf[l] * C.groups;                 // (frame = 8 g + b % 8: one XCD per frame, see the kernel)
      C.band0[l] = bands;
      bands += cdiv(C.oh[l], pp::CH_BAND);
    }
    C.row0[0] = levels[0].row0;
    C.bands_per_frame = bands;
    C.wg0[nlevels] = (int)wg;
    if (wg > 0x3fffffffL) chain = false;
    if (chain) {
      const size_t words = pp::chain_words((size_t)batch, (size_t)bands, (size_t)C.groups);
      bool grew = false;
      if (c->w_chain.ensure(sizeof(uint32_t) * words, &grew) != PISLAM_OK) return fail(c, PISLAM_ERR_NOMEM, "hipMalloc(build chain counters)");
      if (grew || c->chain_words != words) {
        // (a different layout puts [done, fault] elsewhere: start from zeros — stream order puts this behind any launch in flight)
        HIPCHK(c, hipMemsetAsync(c->w_chain.p, 0, c->w_chain.cap, c->stream));
        if (!grew && c->chain_words != 0) c->ovf_layouts++;   // (a captured build holds the old layout: workspace_generation)
        c->chain_words = words;
      }
      PCHK(ensure_fault_flag(c));
      hipLaunchKernelGGL(pp::k_bilinear_chain, dim3((unsigned)wg), dim3(256), 0, c->stream, C, pyramids, pyramid_stride,
                         c->w_chain.as<uint32_t>(), c->frame_flag_dev + 1, (uint32_t)c->opt_frame_test);
      PCHK(launch_ok(c, "k_bilinear_chain"));
    }
  }
  if (!chain)
    for (int l = 0; l + 1 < nlevels; l++) {
      const uint8_t *src = pyramids + (size_t)levels[l].row0 * vstep;
      uint8_t *dst = pyramids + (size_t)levels[l + 1].row0 * vstep;
      const int w = levels[l].width, h = levels[l].height;
      if (steps[l] == 1)
        PCHK((launch_bilinear<8, 7>(c, src, dst, vstep, vstep, pyramid_stride, pyramid_stride, batch, w, h)));
      else
        PCHK((launch_bilinear<16, 13>(c, src, dst, vstep, vstep, pyramid_stride, pyramid_stride, batch, w, h)));
    }
  return PISLAM_OK;
}

// ===========================================================================
// batch pipeline (staged: one launch group per level, blockIdx.z = pyramid)
// ===========================================================================
namespace {
int check_params(pislam_ctx *c, const pislam_frontend_params *p, const pislam_level *lv, int batch) {
  if (!c) return PISLAM_ERR_INVALID;
  if (!p || !lv) return fail(c, PISLAM_ERR_INVALID, "null params");
  if (batch <= 0) return fail(c, PISLAM_ERR_INVALID, "batch must be positive");
  if (p->vstep <= 0 || p->rows <= 0 || p->nlevels < 1 || p->nlevels > 16)
    return fail(c, PISLAM_ERR_INVALID, "bad vstep/rows/nlevels");
  if (p->border < 16) return fail(c, PISLAM_ERR_INVALID, "ORB needs border >= 16 (Fast.h:46-49)");
  if (p->words < 1 || p->words > 8) return fail(c, PISLAM_ERR_INVALID, "words must be 1..8");
  if (p->log_bucket_size < 0 || p->log_bucket_size > 8 || p->bucket_limit < 1 || p->bucket_limit > 64)
    return fail(c, PISLAM_ERR_INVALID, "bad bucket parameters");
  if (p->max_keypoints < 1) return fail(c, PISLAM_ERR_INVALID, "max_keypoints must be positive");
  for (int l = 0; l < p->nlevels; l++) {
    if (lv[l].width <= 0 || lv[l].height <= 0 || lv[l].row0 < 0 || lv[l].col0 < 0 ||
        lv[l].col0 + lv[l].width > p->vstep || lv[l].row0 + lv[l].height > p->rows)
      return fail(c, PISLAM_ERR_INVALID, "level does not fit the pyramid buffer");
    if (lv[l].row0 + lv[l].height > 4096 || lv[l].col0 + lv[l].width > 4096)
      return fail(c, PISLAM_ERR_INVALID, "stacked coordinates exceed 12 bits (Util.h:27-29)");
  }
  return PISLAM_OK;
}
}  // namespace


namespace {

// Strip plan of the fused pipeline.  Strip height per level: aim at ~8k pixels per workgroup,
// even, 16..32 rows (override: option "strip_rows").
bool build_fused_plan_rows(const pislam_ctx *c, const pislam_frontend_params *p, const pislam_level *lv,
                           int batch, int rows_max, pf::FusedParams *F, size_t *lds_bytes, size_t *lds_alias_bytes) {
  if (p->nlevels > pf::MAX_LEVELS) return false;
  memset(F, 0, sizeof(*F));
  F->vstep = p->vstep;
  F->rows = p->rows;
  F->border = p->border;
  F->thr = p->fast_threshold & 0xff;
  F->hthr = p->harris_threshold;
  F->batch = batch;
  F->dump_score = c->opt_dump_score;
  F->lbs = p->log_bucket_size;
  F->limit = p->bucket_limit;
  F->words = p->words;
  F->orb_in_strip = c->opt_orb_in_strip;
  // Buckets (fastExtract<.., logBucketSize, bucketLimit>): by default the strips run exactly as without buckets (F->lbs = 0:
  // any strip height, the plain kernels) and pf::k_bucket_select applies the per-cell top-`limit` between the strip kernel and
  // the gather (run_fused); option "bucket_select" 0 keeps the selection inside the strips (rounds 1-3: strips cut on bucket
  // rows, cells of 4..32 px).
  bool select = p->log_bucket_size != 0 && c->opt_bucket_select;
  for (int l = 0; l < p->nlevels && select; l++)
    if (lv[l].width - 2 * p->border > 0 && ((lv[l].width - 2 * p->border - 1) >> p->log_bucket_size) + 1 > pf::SEL_NB) select = false;
  if (select) {
    F->lbs = 0;
    F->orb_in_strip = 0;
  }
  // fused bucket mode inside the strips: cells of 4..32 px (they must fit a strip and the per-wave scratch)
  if (F->lbs != 0 && (p->log_bucket_size < 2 || p->log_bucket_size > 5)) return false;
  F->ablate = c->opt_ablate;
  int strips = 0, slots = 0, runs = 0;
  size_t lds = 0, lds_alias = 0;
  // ALIAS layout, heuristic strip height for a level under a residency target of `wgs` workgroups per CU:
  // ~16k pixels per strip, 16..28 rows, capped so that queues + tile + the minimum shared queue fit
  // 160 KiB / wgs where 16 rows allow it (VGA at 5 per CU: 24 rows at level 0, 28 below; measured 0.293 ms
  // vs 0.311 ms with 16-row strips).
  // LDS pitch of an image tile row that stages columns [xbase, xend + 8): a multiple of 16 bytes.  (An ODD number of
  // 16-byte vectors — consecutive rows 4 banks apart instead of column x of every row in one bank at VGA level 0's 640
  // bytes — was measured in round 4: bit-exact, the strip kernel within 0.5 % either way: the per-candidate reads'
  // bank conflicts, 45 % of its LDS cycles, come from the candidates' random columns, not from the row pitch.)
  auto tile_pitch = [&](int xend_l) -> int {
    return (xend_l - p->border + ((p->border - 4) & 15) + 4 + 8 + 15) & ~15;
  };
  auto alias_rows = [&](int xend_l, int w, int wgs) -> int {
    const int tpitch_l = tile_pitch(xend_l);
    const long budget = 160 * 1024 / wgs - (long)(pf::WAVES * pf::QCAP + pf::QH_SHARED) * 4;
    const int rcap = (int)(budget / tpitch_l - 10) & ~1;
    if (wgs != 5 && rcap < 10) return 0;              // (an explicit residency request falls back to the generic rule)
    return std::max(16, std::min(std::min(rows_max, std::max(16, (c->opt_strip_px / w) & ~1)), rcap));
  };
  // Residency target of the heuristic: 5 workgroups per CU.  (A search over 5 / 4 / 3 per CU with a cost model
  // "pixels * (R + 4) / R / measured throughput at that residency" was tried for the 1280-wide levels of BASELINE
  // config 4 — 12-row strips at 4 per CU instead of 16 rows at 3 — and measured no better: 1.24 vs 1.21 ms at
  // batch 256; forcing 5 per CU with 8-row strips and no halo carry gave 1.16 ms.  Option "wgs_per_cu" overrides.)
  const int alias_wgs = c->opt_wgs_per_cu > 0 ? c->opt_wgs_per_cu : 5;
  // Plan entries: a level, or the x-tiles of a wide level (pf::FusedLevel).  A level with more than `tile_max`
  // classified columns is cut into tiles of T block-origin columns (T a multiple of 32: tiles start on
  // bucket boundaries for every fused bucket size); tile t > 0 starts HALO = 32 columns to the left of its
  // first block origin, so that, seen as a level of its own with the same border B, it classifies and scores
  // the columns its boundary blocks' NMS reads, and every tile but the last ends 2 columns past its last
  // owned block origin.
  struct Entry {
    int w, h, row0, col0, ex0, ex1, xscore, gfirst, gn, wmax;
  };
  std::vector<Entry> entries;
  {
    const int B = p->border, HALO = 32;
    const int tile_max = c->opt_tile_cols > 0 ? c->opt_tile_cols : (c->opt_tile_cols < 0 ? 1 << 30 : 704);
    for (int l = 0; l < p->nlevels; l++) {
      const int w = lv[l].width, nx = w - 2 * B, ny = lv[l].height - 2 * B;
      int nt = 1;
      // tiles of ~448 owned columns: with the 32-column halo a tile row is two full 256-pixel prefilter steps,
      // and its strips reach the full 28 rows at 5 workgroups per CU (1280x960 batch 256: 0.94 ms with three
      // 416-column tiles at level 0 against 1.01 ms with two of 624)
      if (nx > 0 && ny > 0 && 16 * cdiv(nx, 16) > tile_max)
        nt = std::max(2, tile_max >= 640 ? (nx + 224) / 448 : cdiv(nx, std::max(64, tile_max)));
      const int T = nt > 1 ? (cdiv(nx, nt) + 31) & ~31 : 0;
      nt = nt > 1 ? cdiv(nx, T) : 1;
      const int g0 = (int)entries.size();
      int wmax = 0;
      for (int t = 0; t < nt; t++) {
        Entry e;
        if (nt == 1) {
          e = {w, lv[l].height, lv[l].row0, lv[l].col0, B, w - B, w - B, g0, 1, w};
        } else {
          const int o = t == 0 ? 0 : t * T - HALO;                 // level column of the entry's origin
          const int E = std::min(B + (t + 1) * T, w - B);           // one past the last owned block origin (level x)
          e.w = t == nt - 1 ? w - o : E - o + 2 + B;
          e.h = lv[l].height;
          e.row0 = lv[l].row0;
          e.col0 = lv[l].col0 + o;
          e.ex0 = t == 0 ? B : B + HALO;
          e.ex1 = E - o;
          e.xscore = (w - B) - o;
          e.gfirst = g0;
          e.gn = nt;
        }
        wmax = std::max(wmax, e.w);
        entries.push_back(e);
      }
      for (int t = 0; t < nt; t++) entries[g0 + t].wmax = wmax;
    }
    if ((int)entries.size() > pf::MAX_LEVELS) return false;
  }
  F->nlevels = (int)entries.size();
  for (int l = 0; l < F->nlevels; l++) {
    pf::FusedLevel &L = F->lv[l];
    const Entry &en = entries[l];
    L.w = en.w;
    L.h = en.h;
    L.row0 = en.row0;
    L.col0 = en.col0;
    L.ex0 = en.ex0;
    L.ex1 = en.ex1;
    L.xscore = en.xscore;
    L.gfirst = en.gfirst;
    L.gn = en.gn;
    const int nx = L.w - 2 * p->border, ny = L.h - 2 * p->border;
    L.strip0 = strips;
    L.slot0 = slots;
    if (nx <= 0 || ny <= 0) {   // nothing to extract on this level (Fast.h loops do not run)
      L.R = 16;
      L.nstrips = 0;
      L.nbx = 0;
      L.xend = p->border;
      L.pitch = 16;
      continue;
    }
    // (the strip height comes from the widest tile of the level: all its tiles must cut the same strips)
    const int nx_r = en.wmax - 2 * p->border;
    const int xend_l = p->border + 16 * cdiv(nx_r, 16), pitch_l = (xend_l + 4 + 15) & ~15;
    const size_t qbytes = (pf::WAVES * pf::QCAP + pf::SHARED_Q) * sizeof(uint32_t);
    int R = c->opt_strip_rows;
    if (R == 0 && c->opt_alias) R = alias_rows(xend_l, en.wmax, alias_wgs);
    if (R == 0) {
      R = (8192 / en.wmax) & ~1;         // ~8k pixels per strip ...
      R = std::min(32, std::max(16, R));
      if (c->opt_wgs_per_cu > 0) {       // ... capped so that tiles + queues fit 160 KiB / wgs_per_cu
        const size_t budget = (size_t)(160 * 1024) / (size_t)c->opt_wgs_per_cu;
        if (budget > qbytes + 13 * (size_t)pitch_l) {
          const int rcap = (int)(((budget - qbytes) / pitch_l - 13) / 2) & ~1;
          R = std::max(8, std::min(R, rcap));
        }
      }
    }
    if (F->lbs) {                        // (selection inside the strips:) strips hold whole bucket rows
      const int bs = 1 << p->log_bucket_size;
      if (c->opt_strip_rows == 0 && c->opt_alias) {
        // heuristic height: round UP to whole bucket rows (16-px buckets: 32-row strips — measured 0.315 ms
        // vs 0.371 ms with 16-row strips) where that still fits the residency budget, DOWN on the levels where it
        // does not: the launch has ONE LDS size, a single level over the budget costs every level its fifth workgroup
        const int up = std::min(std::max(bs, 32), ((R + bs - 1) / bs) * bs), down = std::max(bs, (R / bs) * bs);
        const int tp = tile_pitch(xend_l);
        const long need = (long)(pf::WAVES * pf::QCAP + pf::QH_SHARED) * 4 + (long)(up + 10) * tp;
        R = (c->opt_bucket_round_up || need <= 160 * 1024 / alias_wgs - 1280) ? up : down;
      } else
        R = std::max(bs, (R / bs) * bs);
    }
    L.R = R;
    L.nstrips = cdiv(ny, R);
    L.nbx = (nx + 1) / 2;
    L.xend = p->border + 16 * cdiv(nx, 16);
    L.pitch = (L.xend + 4 + 15) & ~15;
    {
      // staged columns [xbase, xbase+tpitch) with xbase = (border-4) & ~15 must reach column xend+8
      L.tpitch = tile_pitch(L.xend);
    }
    L.vpr_recip = (uint32_t)(((1ull << 32) + (L.tpitch / 16) - 1) / (L.tpitch / 16));
    L.tp_recip = (uint32_t)(((1ull << 32) + L.tpitch - 1) / L.tpitch);
    strips += L.nstrips;
    slots += L.nstrips * (R / 2) * L.nbx;
    // scan fallbacks reuse the image tile: row buffers of R/2 x nbx dwords, or per-cell results (<= one
    // dword per block) + per-cell counts (<= a quarter of that: a cell holds >= 2x2 blocks)
    L.tbytes = std::max((R + 10) * L.tpitch, (((R / 2) * L.nbx * 5 + 64) + 15) & ~15);
    lds = std::max(lds, (size_t)L.tbytes + (size_t)(R + 3) * L.pitch +
                            (pf::WAVES * pf::QCAP + pf::SHARED_Q) * sizeof(uint32_t));
    {
      // ALIAS layout: the score tile (R+3 rows) is laid over [NMS scratch end, image row R): pad the
      // per-wave queue area when that span is too short (levels wider than ~680 columns)
      // ... and the ORB phase's 8 patches + vrecpe table are laid over the same span (strip_body phase E)
      const long need = std::max((long)pf::NMS_SCRATCH * 4 + (long)(R + 3) * L.pitch,
                                 (long)pf::NMS_SCRATCH * 4 + (long)pf::WAVES * 2 * pf::ORB_PATCH_BYTES + 256);
      const long have = (long)pf::WAVES * pf::QCAP * 4 + (long)R * L.tpitch;
      L.apad = need > have ? (int)((need - have + 15) & ~15L) : 0;
      // one shared queue in this layout: at least QH_SHARED entries, and whatever LDS the level's tile
      // leaves under the 5-workgroups-per-CU budget (narrow levels of a textured photo are the dense ones)
      const long fixed = (long)pf::WAVES * pf::QCAP * 4 + L.apad + (long)(R + 10) * L.tpitch;
      // (margin of 1280 B: static LDS + allocation granule — with 512 B the kernel measurably lost its 5th workgroup)
      const long spare = (160 * 1024 / alias_wgs - 1280 - fixed) / 4;
      L.qh = (int)std::min<long>(4096, std::max<long>(pf::QH_SHARED, spare & ~3L));
      lds_alias = std::max(lds_alias, (size_t)fixed + (size_t)L.qh * 4);
    }
  }
  // Runs: a workgroup walks run_len consecutive strips of a level (halo carried in LDS).  Longer runs save the
  // duplicated halo work but leave fewer, longer workgroups for the dispatcher to balance over the 5 resident
  // slots per CU.  Measured (strip kernel, ms): VGA batch 256 (15 strips per slot): 0.236 / 0.231 / 0.233 / 0.243 /
  // 0.242 / 0.235 / 0.231 / 0.229 / 0.229 / 0.243 for run_len 1 / 2 / 3 / 4 / 5 / 6 / 8 / 10 / 12 / 16 — the carry is
  // worth ~3 % at best and the curve is dispatch-quantisation noise; 720p batch 64 (10 strips per slot): 0.208 /
  // 0.207 / 0.216 / 0.227 / 0.238 for 1..5; 1280x960 batch 256 (54 per slot): 1.029 / 1.018 / 1.012 / 1.044 / 1.053 /
  // 1.010 for 4 / 5 / 6 / 7 / 8..10 / 12, batch 64: 0.278 / 0.307 / 0.376 for 2 / 3 / 8; VGA batch 32, 64: 1 is best.
  // Rule: ~6.5 workgroups per resident slot, at most 8 strips per run.  (List-scheduling and processor-sharing
  // simulations of one XCD were tried as a predictor: neither tracks the measured 2-3 % structure.)
  // With the runs of a pyramid launched longest first (order[], below) the curve flattens: VGA batch 256
  // 0.232 / 0.222 / 0.220 / 0.224 / 0.222 / 0.233 for run_len 1 / 2 / 3 / 4 / 6 / 8; 1280x960 batch 256 0.920 / 0.898 /
  // 0.899 / 0.889 / 0.889 for 2 / 4 / 6 / 8 / 12; 720p batch 64 0.191 / 0.191 / 0.197 / 0.206 for 1 / 2 / 3 / 4.
  {
    // Round 5 (strips of a run follow each other without a barrier): one call at a time the rule stands (VGA batch 256: strip
    // kernel 0.164 / 0.166 / 0.171 ms for run_len 2 / 3 / 4), but as a LANE of a pipeline — other batches' kernels fill the
    // tail of the launch — longer runs win: whole step 0.2233 / 0.2208 / 0.2203 ms for 2 / 3 / 4 (720p batch 64: 0.2974 /
    // 0.2954 for 2 / 3; demo photo x 256: 0.3242 / 0.3208 for 2 / 4).  Lanes aim at ~3.25 workgroups per slot.
    const double per_slot = (double)strips * batch / (5.0 * std::max(1, c->num_cus));
    const double per_wg = c->lanes_in_flight > 1 ? 3.25 : 6.5;
    F->run_len = c->opt_run_len > 0 ? c->opt_run_len : std::max(1, std::min(8, (int)(per_slot / per_wg + 0.5)));
    // Round 6, lanes only: the LONGEST runs that still leave the launch 0.6 workgroups per resident slot (5 per CU) — whole
    // levels per workgroup where the batch is large enough.  Re-swept after the pretest change, pipelined step in ms for
    // run_len default (the rule above) / 12 / 16 / 24 / 32 / 64: VGA batch 256 0.2117 / 0.2087 / 0.2062 / 0.2072 / 0.2063 / 0.2068
    // (8 whole-level runs per pyramid = 1.6 per slot); demo photo 0.3169 / 0.3153 / 0.3153 / 0.3144 / 0.3143 / 0.3140; 1280x960
    // 0.6692 / 0.6674 / 0.6628 / 0.6637 / 0.6517 / 0.6531; 720p build batch 64 0.2865 / 0.2759 / 0.2671 / 0.2581 / 0.2589 / 0.2590
    // (15 runs per pyramid = 0.75 per slot); bucket mode 0.2334 / 0.2293 / 0.2285 / 0.2281 / 0.2287 / 0.2277 — and where the
    // launch gets too few workgroups it turns: VGA batch 64 0.0646 / 0.0631 (0.6 per slot) / 0.0681 (0.5) / 0.0720; batch 16
    // 0.0208 / 0.0425.  The strip kernel ALONE is slower with long runs (0.156 -> 0.159-0.162 ms: a longer tail); on a lane
    // another batch's kernels run in that tail, and a long run stages its halo once and prefetches every strip but the first.
    if (c->opt_run_len <= 0 && c->lanes_in_flight > 1) {
      const long want = (long)(0.6 * 5.0 * std::max(1, c->num_cus));
      int longest = 1;
      for (int l = 0; l < F->nlevels; l++) longest = std::max(longest, F->lv[l].nstrips);
      longest = std::min(longest, 64);                 // (option range; k_frame keeps a 64-bit mask of a run's strips)
      for (int rl = longest; rl > F->run_len; rl--) {
        long r = 0;
        for (int l = 0; l < F->nlevels; l++) r += cdiv(F->lv[l].nstrips, rl);
        if (r * batch >= want && r <= pf::MAX_ORDER) {
          F->run_len = rl;
          break;
        }
      }
    }
  }
  for (int l = 0; l < F->nlevels; l++) {
    F->lv[l].run0 = runs;
    F->lv[l].nruns = cdiv(F->lv[l].nstrips, F->run_len);
    runs += F->lv[l].nruns;
  }
  // Launch order of a pyramid's runs: estimated cost (pixels + a fixed share per strip), longest first — the
  // longest-processing-time-first rule of list scheduling: workgroups are dispatched in blockIdx order to the
  // 5 resident slots per CU, so the short runs of the small levels fill the tail of the launch.  Entry order
  // (level by level) interleaves short last-runs of big levels with long runs of the next level: VGA batch 256
  // 0.232 -> 0.222 ms, 1280x960 batch 256 0.927 -> 0.895 ms.
  F->order_n = 0;
  if (runs > 0 && runs <= pf::MAX_ORDER && c->opt_run_order) {
    struct RunCost {
      double cost;
      int entry, run;
    };
    std::vector<RunCost> rc;
    for (int l = 0; l < F->nlevels; l++) {
      const pf::FusedLevel &L = F->lv[l];
      const int ny = L.h - 2 * p->border;
      for (int r = 0; r < L.nruns; r++) {
        double cst = 0;
        for (int sidx = r * F->run_len; sidx < std::min((r + 1) * F->run_len, L.nstrips); sidx++)
          cst += 1.6 * L.w * std::min(L.R, ny - sidx * L.R) + 6000.0;
        rc.push_back({cst, l, r});
      }
    }
    std::stable_sort(rc.begin(), rc.end(), [](const RunCost &a, const RunCost &b) { return a.cost > b.cost; });
    for (size_t i = 0; i < rc.size(); i++) F->order[i] = ((uint32_t)rc[i].entry << 16) | (uint32_t)rc[i].run;
    F->order_n = (int)rc.size();
  }
  F->strips_per_pyr = strips;
  F->runs_per_pyr = runs;
  F->slots_per_pyr = slots;
  *lds_bytes = lds;
  *lds_alias_bytes = lds_alias + (size_t)c->opt_lds_pad;   // profiling: opt_lds_pad lowers the residency artificially
  if ((size_t)p->rows * p->vstep > 0x7fffffffu) return false;   // 32-bit byte offsets inside a pyramid
  return lds <= 150 * 1024;       // (strips == 0: no level holds a classifiable pixel — the caller writes zero counts)
}

// Cap of the heuristic strip height.  Every strip pays a fixed share (set-up, barriers, the halo carried through
// LDS, the ragged last wave step of each phase), so the narrow levels want strips as tall as the prefetch
// registers allow (R * pitch <= 16 KiB) — but only when the launch has plenty of workgroups per resident slot;
// a small launch is balanced better by more, shorter strips.  Measured, strip kernel alone (ms) for caps 28 / 36 /
// 44 / 56 / 64: VGA batch 256 (15 strips per slot at cap 28) 0.221 / 0.212 / 0.209 / 0.207 / 0.211; 1280x960 batch
// 256 (54 per slot) 0.885 / 0.857 / 0.861 / 0.861; 720p batch 64 (10 per slot): whole step 0.337 / 0.346 / 0.346 /
// 0.344.  Rule: cap 56 from 12 strips per slot on, 28 below.  Option "strip_rows_max" overrides.
bool build_fused_plan(const pislam_ctx *c, const pislam_frontend_params *p, const pislam_level *lv, int batch,
                      pf::FusedParams *F, size_t *lds_bytes, size_t *lds_alias_bytes) {
  if (c->opt_strip_rows_max > 0)
    return build_fused_plan_rows(c, p, lv, batch, c->opt_strip_rows_max, F, lds_bytes, lds_alias_bytes);
  if (!build_fused_plan_rows(c, p, lv, batch, 28, F, lds_bytes, lds_alias_bytes)) return false;
  const double per_slot = (double)F->strips_per_pyr * batch / (5.0 * std::max(1, c->num_cus));
  if (per_slot < 12.0) return true;
  pf::FusedParams tall;
  size_t l0 = 0, l1 = 0;
  if (build_fused_plan_rows(c, p, lv, batch, 56, &tall, &l0, &l1)) {
    *F = tall;
    *lds_bytes = l0;
    *lds_alias_bytes = l1;
  }
  return true;
}

// Sub-batches of a fused batch call (option "sub_batches", default 1 = one launch group): the strip kernels of the
// sub-batches run back to back on the context stream, overflow pass + gather/ORB of sub-batch i on the context's
// second stream under the strip kernel of sub-batch i+1 (fork / join by events inside the call).  MEASURED AND
// NOT THE DEFAULT (MI355X, VGA batch 256, one call at a time): 0.286 ms with one launch group, 0.326 / 0.360 / 0.367
// / 0.466 ms with 2 / 3 / 4 / 6 sub-batches — a strip launch of 128 pyramids takes 0.121 ms, not half of 0.206: every
// launch ends in its own tail of partly filled CUs and the serialised launches add ~18 us each, more than the
// overlap wins back (1280x960: 1.23 -> 1.36 ms with 4).  What does pay is whole batches in flight on separate
// contexts (bench.py --streams, tools/pislam_demo --streams: 0.252 ms).  `0` = the MiB-per-sub-batch rule below.
int choose_sub_batches(const pislam_ctx *c, const pislam_frontend_params *p, int batch) {
  if (c->opt_dump_score || c->opt_ablate || c->opt_pipeline == 1) return 1;
  int n = c->opt_sub_batches;
  if (n == 0) {
    const double mb = (double)p->rows * p->vstep * batch / (1024.0 * 1024.0);
    const double target = c->opt_sub_mb > 0 ? c->opt_sub_mb : 128.0;
    n = (int)(mb / target + 0.5);
    n = std::min(n, batch / 16);
  }
  return std::max(1, std::min(std::min(n, batch), (int)pislam_ctx::MAX_SUB));
}
inline int sub_max(int batch, int nsub) { return batch / nsub + (batch % nsub ? 1 : 0); }

// Overflow lists: one per sub-batch ([0] count, [1] count of the previous step, [2..] entries), `stride` dwords
// apart.  A new layout (or a new allocation) starts from an all-zero buffer: a stale entry must never be read
// as a list header.
int prepare_ovf(pislam_ctx *c, int nsub, size_t stride) {
  bool grew = false;
  if (c->w_ovf.ensure(sizeof(uint32_t) * stride * nsub, &grew) != PISLAM_OK)
    return fail(c, PISLAM_ERR_NOMEM, "hipMalloc(overflow list)");
  // (one list: its header is at offset 0 whatever the stride — calls of different batch sizes share the layout)
  const size_t lay = nsub == 1 ? 0 : stride;
  if (grew || c->ovf_nsub != nsub || c->ovf_stride != lay) {
    HIPCHK(c, hipMemsetAsync(c->w_ovf.p, 0, c->w_ovf.cap, c->stream));
    c->ovf_nsub = nsub;
    c->ovf_stride = lay;
    c->ovf_layouts++;
  }
  return PISLAM_OK;
}

int ensure_aux(pislam_ctx *c, int nsub) {
  if (!c->aux_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
  if (!c->ev_join) HIPCHK(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  for (int i = 0; i < nsub; i++)
    if (!c->ev_sub[i]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_sub[i], hipEventDisableTiming));
  return PISLAM_OK;
}

// The bucket selection pass (pf::k_bucket_select) and the UNIT plan the gather runs on when the strips run as without buckets
// (build_fused_plan_rows): one "strip" per (level, cell row), `buckets x limit` slots each, lists final (lbs != 0, no tiles:
// the gather concatenates).  Shared by run_fused and pislam_frontend_reserve (which sizes w_ustage / w_ucount from it, so that
// the first bucket-mode call after a reserve allocates nothing and can be captured into a hipGraph).
int build_select_plan(pislam_ctx *c, const pislam_frontend_params *p, const pf::FusedParams &Fplan, pf::SelectPlan *Qp,
                      pf::FusedParams *Up) {
  pf::SelectPlan &Q = *Qp;
  pf::FusedParams &U = *Up;
  memset(&Q, 0, sizeof(Q));
  memset(&U, 0, sizeof(U));
  const int lbs = p->log_bucket_size, bs = 1 << lbs, B = p->border;
  Q.lbs = lbs;
  Q.limit = p->bucket_limit;
  Q.border = B;
  int nreal = 0;
  for (int e = 0; e < Fplan.nlevels; e++) {
    if (Fplan.lv[e].gfirst != e) continue;           // (the tiles of a level follow its first entry)
    if (nreal >= 16) return fail(c, PISLAM_ERR_INVALID, "too many levels for the bucket selection pass");
    const int l = nreal++;
    const int gn = std::max(1, Fplan.lv[e].gn);
    const pf::FusedLevel &last = Fplan.lv[e + gn - 1];
    const int wl = last.col0 + last.w - Fplan.lv[e].col0, hl = Fplan.lv[e].h;       // the level's own size
    const int nx = wl - 2 * B, ny = hl - 2 * B;
    Q.row0[l] = Fplan.lv[e].row0;
    Q.col0[l] = Fplan.lv[e].col0;
    Q.h[l] = hl;
    Q.g0[l] = e;
    Q.gn[l] = gn;
    Q.unit0[l] = Q.units_per_pyr;
    Q.nunits[l] = (nx > 0 && ny > 0) ? cdiv(ny, bs) : 0;
    Q.cap[l] = (nx > 0 ? ((nx - 1) >> lbs) + 1 : 1) * p->bucket_limit;             // Fast.h:201 numBuckets x bucketLimit
    Q.uslot0[l] = Q.uslots_per_pyr;
    Q.nb_max = std::max(Q.nb_max, Q.cap[l] / p->bucket_limit);
    Q.units_per_pyr += Q.nunits[l];
    Q.uslots_per_pyr += Q.nunits[l] * Q.cap[l];
    pf::FusedLevel &ul = U.lv[l];
    ul.w = wl;
    ul.h = hl;
    ul.row0 = Q.row0[l];
    ul.col0 = Q.col0[l];
    ul.R = 2;                                        // strip_slot_of: slot0 + s * (R >> 1) * nbx
    ul.nbx = Q.cap[l];
    ul.nstrips = Q.nunits[l];
    ul.strip0 = Q.unit0[l];
    ul.slot0 = Q.uslot0[l];
    ul.gfirst = l;
    ul.gn = 1;
  }
  Q.nlevels = nreal;
  U.nlevels = nreal;
  U.strips_per_pyr = Q.units_per_pyr;
  U.slots_per_pyr = Q.uslots_per_pyr;
  U.vstep = p->vstep;
  U.rows = p->rows;
  U.border = B;
  U.lbs = lbs;
  U.limit = p->bucket_limit;
  U.words = p->words;
  U.ablate = Fplan.ablate;
  if (Q.units_per_pyr == 0) return fail(c, PISLAM_ERR_INVALID, "no extractable level");
  return PISLAM_OK;
}

// The device copy of a host-built plan table: found by content, or created (hipMalloc + upload in stream order — never
// inside a capture: the first occurrence of a call runs eagerly, and pislam_frontend_reserve builds the tables).
int plan_table(pislam_ctx *c, std::vector<uint32_t> &&t, const uint32_t **out) {
  if (t.empty()) t.push_back(0u);
  for (size_t i = 0; i < c->plan_tables.size(); i++)
    if (c->plan_tables[i]->host == t) {
      std::rotate(c->plan_tables.begin() + i, c->plan_tables.begin() + i + 1, c->plan_tables.end());   // most recently used last, the others keep their order
      *out = c->plan_tables.back()->dev.as<uint32_t>();
      return PISLAM_OK;
    }
  if (c->plan_tables.size() >= 64) {                  // evict the least recently used (a table is a few KB)
    HIPCHK(c, hipStreamSynchronize(c->stream));       // (a launch in flight may still read it)
    c->plan_tables.front()->dev.release();
    delete c->plan_tables.front();
    c->plan_tables.erase(c->plan_tables.begin());
    c->table_uploads++;
  }
  pislam_ctx::PlanTable *pt = new pislam_ctx::PlanTable();
  pt->host = std::move(t);
  if (pt->dev.ensure(sizeof(uint32_t) * pt->host.size()) != PISLAM_OK) {
    delete pt;
    return fail(c, PISLAM_ERR_NOMEM, "hipMalloc(plan table)");
  }
  c->plan_tables.push_back(pt);
  HIPCHK(c, hipMemcpyAsync(pt->dev.p, pt->host.data(), sizeof(uint32_t) * pt->host.size(), hipMemcpyHostToDevice, c->stream));
  *out = pt->dev.as<uint32_t>();
  return PISLAM_OK;
}

// The unit table of the bucket selection pass (pf::k_bucket_select reads one record per unit instead of walking the plan):
// built on the host from the strip plan and the selection plan; its device copy comes from plan_table().
int ensure_unit_table(pislam_ctx *c, const pislam_frontend_params *p, const pf::FusedParams &F, const pf::SelectPlan &Q) {
  std::vector<uint32_t> t((size_t)Q.units_per_pyr * pf::SEL_REC, 0u);
  const int B = p->border, lbs = p->log_bucket_size, bs = 1 << lbs;
  for (int l = 0; l < Q.nlevels; l++)
    for (int cr = 0; cr < Q.nunits[l]; cr++) {
      uint32_t *r = &t[(size_t)(Q.unit0[l] + cr) * pf::SEL_REC];
      const int y0 = B + (cr << lbs), y1 = std::min(y0 + bs, Q.h[l] - B);
      int nl = 0;
      for (int k = 0; k < Q.gn[l]; k++) {
        const pf::FusedLevel &E = F.lv[Q.g0[l] + k];
        const int s_lo = (y0 - B) / E.R, s_hi = std::min((y1 - 1 - B) / E.R, E.nstrips - 1);
        for (int sidx = s_lo; sidx <= s_hi; sidx++, nl++)
          if (nl < pf::SEL_ML) {
            r[8 + nl] = (uint32_t)(E.slot0 + sidx * (E.R >> 1) * E.nbx);
            r[16 + nl] = (uint32_t)(E.strip0 + sidx);
          }
      }
      r[0] = nl <= pf::SEL_ML ? (uint32_t)nl : 0xffffffffu;
      r[1] = (uint32_t)l;
      r[2] = (uint32_t)cr;
      r[3] = (uint32_t)(Q.uslot0[l] + cr * Q.cap[l]);
      r[4] = (uint32_t)(Q.row0[l] + B);
      r[5] = (uint32_t)(Q.col0[l] + B);
    }
  return plan_table(c, std::move(t), &c->cur_utab);
}

// The one-launch path (pf::k_frame) takes batches of up to FRAME_MAX_BATCH pyramids: three launch floors are most of such
// a call (one VGA pyramid: 31 us in three launches, 15 us of it work), and its gather + ORB workgroups — which wait inside
// the grid for their pyramid's strips — stay a small fraction of an XCD's resident slots even with several such launches
// in flight (at most 128 per launch).
constexpr int FRAME_MAX_BATCH = 8;                                  // what option "frame" can be raised to
constexpr int FRAME_DEFAULT_BATCH = 2;                              // measured: one launch wins for 1 and 2 pyramids per call
inline int frame_max_batch(const pislam_ctx *c) { return c->opt_frame <= 0 ? 0 : c->opt_frame == 1 ? FRAME_DEFAULT_BATCH : std::min(c->opt_frame, FRAME_MAX_BATCH); }
inline int frame_chunks(int batch) { return std::min(64, std::max(16, 128 / std::max(1, batch))); }   // ORB workgroups per pyramid
static_assert(FRAME_MAX_BATCH <= pf::FRAME_SYNC_PYR, "k_frame's hand-over counters");
int ensure_frame_sync(pislam_ctx *c) {
  bool grew = false;
  if (c->w_sync.ensure(sizeof(uint32_t) * pf::FRAME_SYNC_WORDS, &grew) != PISLAM_OK)
    return fail(c, PISLAM_ERR_NOMEM, "hipMalloc(frame sync)");
  if (grew) HIPCHK(c, hipMemsetAsync(c->w_sync.p, 0, c->w_sync.cap, c->stream));   // (the kernel re-arms them itself)
  return ensure_fault_flag(c);
}
// `Fplan`: the strip plan, built for the largest sub-batch (sub_max pyramids).
int run_fused(pislam_ctx *c, const pislam_frontend_params *p, const pf::FusedParams &Fplan, size_t lds, size_t lds_alias,
              const uint8_t *pyramids, size_t stride, int batch, int nsub, uint32_t *kp, uint32_t *desc, uint32_t *counts) {
  const int S = Fplan.strips_per_pyr;
  const int submax = sub_max(batch, nsub);
  // Buckets with the strips run as without (build_fused_plan_rows): the selection pass and the UNIT plan the gather runs on —
  // one "strip" per (level, cell row), `buckets x limit` slots each, lists final (lbs != 0, no tiles: the gather concatenates).
  const bool sel = p->log_bucket_size != 0 && Fplan.lbs == 0;
  pf::SelectPlan Q;
  pf::FusedParams U;
  memset(&Q, 0, sizeof(Q));
  memset(&U, 0, sizeof(U));
  if (sel) {
    PCHK(build_select_plan(c, p, Fplan, &Q, &U));
    if (c->w_ustage.ensure(sizeof(uint32_t) * (size_t)Q.uslots_per_pyr * batch) != PISLAM_OK ||
        c->w_ucount.ensure(sizeof(uint32_t) * (size_t)Q.units_per_pyr * batch) != PISLAM_OK)
      return fail(c, PISLAM_ERR_NOMEM, "hipMalloc(bucket selection staging)");
    PCHK(ensure_unit_table(c, p, Fplan, Q));
  }
  c->last_path = PISLAM_PATH_FUSED | (sel ? PISLAM_PATH_BUCKET_SELECT : 0u) | (Fplan.lbs != 0 ? PISLAM_PATH_BUCKETS_IN_STRIPS : 0u);
  const int Sg = sel ? Q.units_per_pyr : S;            // "strips" of the plan the gather runs on
  // descriptor staging: QS_SHARED slots of `words` dwords per strip (ALIAS strips hold at most QS_SHARED survivors)
  // (only strips that describe their own keypoints write there: option "orb_in_strip")
  const size_t sdesc_per_pyr = Fplan.orb_in_strip ? (size_t)S * pf::QS_SHARED * (size_t)p->words : 0;
  // (+ 64 dwords: pf::k_bucket_select requests the first 64 slots of a strip's list whatever its count)
  if (c->w_stage.ensure(sizeof(uint32_t) * ((size_t)Fplan.slots_per_pyr * batch + 64)) != PISLAM_OK ||
      c->w_stripcnt.ensure(sizeof(uint32_t) * (size_t)S * batch) != PISLAM_OK ||
      c->w_stagedesc.ensure(sizeof(uint32_t) * sdesc_per_pyr * batch) != PISLAM_OK)
    return fail(c, PISLAM_ERR_NOMEM, "hipMalloc(fused staging)");
  // 16-byte loads need 16-byte aligned rows
  bool vec = ((uintptr_t)pyramids % 16 == 0) && (stride % 16 == 0) && (p->vstep % 16 == 0);
  for (int l = 0; l < Fplan.nlevels; l++) vec = vec && (Fplan.lv[l].col0 % 16 == 0);
  uint8_t *dump = Fplan.dump_score ? c->w_score.as<uint8_t>() : nullptr;
  const size_t dump_stride = (size_t)p->rows * p->vstep;
  // ALIAS layout (score tile laid over the dead image rows, 26 KB instead of 39 KB of LDS per workgroup
  // at VGA): the default.  Its overflow list (strips with overflowing queues) is drained by
  // k_fused_overflow right after; the gather kernel empties the list for the next step.
  const bool alias = c->opt_alias && submax <= 65535 && S <= 65535;
  const size_t ovf_stride = 2 + (size_t)S * submax;
  if (alias) {
    PCHK(prepare_ovf(c, nsub, ovf_stride));
    c->last_strips = (uint32_t)S * (uint32_t)batch;
  }
  if (nsub > 1) PCHK(ensure_aux(c, nsub));
  // HOOKS instantiations: score-map dump (debug / parity hook) and the profiling ablations
  const bool hooks = Fplan.dump_score || Fplan.ablate;
  using KernT = void (*)(const pf::FusedParams, const uint8_t *, size_t, uint32_t *, uint32_t *, uint8_t *, size_t,
                         unsigned long long *, uint32_t *, uint32_t *);
  static const KernT kerns[8] = {
      pf::k_fused_strips<false, false, false>, pf::k_fused_strips<false, false, true>,
      pf::k_fused_strips<false, true, false>,  pf::k_fused_strips<false, true, true>,
      pf::k_fused_strips<true, false, false>,  pf::k_fused_strips<true, false, true>,
      pf::k_fused_strips<true, true, false>,   pf::k_fused_strips<true, true, true>};
  // strips describing their own keypoints (option "orb_in_strip"): separate instantiations of the aligned ALIAS kernels
  static const KernT kerns_orb[2] = {pf::k_fused_strips<true, false, true, true>, pf::k_fused_strips<true, true, true, true>};
  // the default mode's kernels (aligned ALIAS layout, no buckets, gather+ORB describes): compiled without the bucket code
  static const KernT kerns_nb[2] = {pf::k_fused_strips<true, false, true, false, false>,
                                    pf::k_fused_strips<true, true, true, false, false>};
  const KernT kern = (Fplan.orb_in_strip && vec && alias) ? kerns_orb[hooks ? 1 : 0]
                     : (vec && alias && Fplan.lbs == 0)   ? kerns_nb[hooks ? 1 : 0]
                                                          : kerns[(vec ? 4 : 0) | (hooks ? 2 : 0) | (alias ? 1 : 0)];
  const size_t klds = alias ? lds_alias : lds;
  if (klds > 150 * 1024) return fail(c, PISLAM_ERR_INVALID, "level too wide for the strip kernel's LDS tiles");
  if (klds > 64 * 1024)
    HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)klds));
  using OvfT = void (*)(const pf::FusedParams, const uint8_t *, size_t, uint32_t *, uint32_t *, uint8_t *, size_t,
                        const uint32_t *);
  static const OvfT okerns[4] = {pf::k_fused_overflow<false, false>, pf::k_fused_overflow<false, true>,
                                 pf::k_fused_overflow<true, false>, pf::k_fused_overflow<true, true>};
  const OvfT okern = okerns[(vec ? 2 : 0) | (hooks ? 1 : 0)];
  if (alias && lds > 64 * 1024)
    HIPCHK(c, hipFuncSetAttribute((const void *)okern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // k_gather_orb's 48-byte row windows assume a row-independent byte shift (vstep % 16 == 0) and
  // 32-bit byte offsets inside a pyramid; other layouts take the generic gather + per-keypoint ORB kernels.
  const bool generic_orb = p->vstep % 16 != 0 || (size_t)p->rows * p->vstep > 0x7fffffffu;
  // gather + orbCompute in one launch: (chunks, pyramids) workgroups
  // Workgroups per pyramid: one per ~70 k classified pixels (keypoint counts are not known to the host; ~70-100
  // keypoints per workgroup measured best: VGA, 981 keypoints: 14 = 21 chunks > 7, 28; 1280x960, 4389 keypoints: 42-49
  // chunks 0.36 ms against 0.41 ms with 14), at least 16 for small batches, and such that the grid is a whole number
  // of "waves" of resident workgroups (7 per CU: 64 VGPRs, 13.5 KB LDS) — batch 256: 14 chunks = 2 x 1792 workgroups
  // measured 0.079 ms against 0.085 ms with 16 (2.3 waves: the last one 30 % full).
  int nch = 0;
  size_t per_max = 0, olds = 0;
  // (the profiling instantiation exists for the gather's per-phase counters: option "ablate" bits 20..23, pf::orb_describe)
  const auto gkern = (Fplan.ablate >> 20) & 15 ? pf::k_gather_orb<true> : pf::k_gather_orb<false>;
  if (!generic_orb) {
    long px = 0;
    for (int l = 0; l < Fplan.nlevels; l++) px += (long)(Fplan.lv[l].ex1 - Fplan.lv[l].ex0) * Fplan.lv[l].nstrips * Fplan.lv[l].R;
    const int by_px = (int)(px / 70000);
    nch = std::min(64, submax >= 128 ? std::max(8, by_px) : std::max(by_px, std::max(16, 4096 / submax)));
    if (nsub == 1) {
      const long slots = 7L * std::max(1, c->num_cus);
      long best = nch, bestd = 1L << 40;
      for (long m = 1; m <= 64; m++) {
        const long cand = m * slots / submax;
        if (cand < 4 || cand > 64) continue;
        const long d = std::labs(cand - nch);
        if (d < bestd) {
          bestd = d;
          best = cand;
        }
      }
      nch = (int)best;
    } else {
      // pipelined sub-batches: the kernel shares the GPU with the next sub-batch's strip kernel, whole "waves" of
      // resident workgroups mean nothing there — one workgroup per ~70 k pixels
      nch = std::min(64, std::max(8, by_px));
    }
    if (c->opt_orb_chunks > 0) nch = c->opt_orb_chunks;
    per_max = ((size_t)p->max_keypoints + nch - 1) / nch;
    olds = pf::orb_lds_bytes(Sg, per_max);
    if (olds > 150 * 1024) return fail(c, PISLAM_ERR_INVALID, "max_keypoints too large for the fused ORB kernel");
    if (olds > 64 * 1024)
      HIPCHK(c, hipFuncSetAttribute((const void *)gkern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)olds));
  }

  // ---- small batches: strips -> (overflowed strips redone in place) -> gather + ORB as ONE launch ----
  if (c->frame_disabled) c->last_path |= PISLAM_PATH_FRAME_TIMED_OUT;
  if (alias && vec && !sel && Fplan.lbs == 0 && !hooks && !generic_orb && nsub == 1 && !Fplan.orb_in_strip &&
      c->opt_repeat_strips <= 1 && batch <= frame_max_batch(c) && !c->frame_disabled) {
    const int fch = c->opt_orb_chunks > 0 ? std::min(c->opt_orb_chunks, 128) : frame_chunks(batch);
    const size_t fper = ((size_t)p->max_keypoints + fch - 1) / fch;
    const size_t flds = std::max(std::max(lds_alias, lds), pf::orb_lds_bytes(S, fper));
    // Occupancy gate: the gather + ORB workgroups WAIT inside the grid.  With `lanes_in_flight` such launches on the device
    // at most lanes x batch x fch of them are resident, an eighth per XCD; they must stay a small fraction (a quarter) of
    // the workgroups an XCD holds with this launch's LDS footprint (96 VGPRs: five 4-wave workgroups per CU at most), so
    // that strip workgroups always find a slot.
    const long wg_per_cu = std::max<long>(1, std::min<long>(5, (long)(160 * 1024) / (long)std::max<size_t>(flds, 1)));
    const long slots_per_xcd = wg_per_cu * std::max(1, c->num_cus / 8);
    const long waiting_per_xcd = ((long)std::max(1, c->lanes_in_flight) * batch * fch + 7) / 8;
    if (flds <= 150 * 1024 && 4 * waiting_per_xcd <= slots_per_xcd) {
      PCHK(ensure_frame_sync(c));
      if (flds > 64 * 1024)
        HIPCHK(c, hipFuncSetAttribute((const void *)pf::k_frame, hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds));
      pf::FusedParams F = Fplan;
      F.batch = batch;
      const unsigned grid = (unsigned)(batch * F.runs_per_pyr + batch * fch);
      hipLaunchKernelGGL(pf::k_frame, dim3(grid), dim3(pf::NT), flds, c->stream, F, pyramids, stride, c->w_stage.as<uint32_t>(),
                         c->w_stripcnt.as<uint32_t>(), kp, (size_t)p->max_keypoints, (uint32_t)p->max_keypoints, counts, desc,
                         (size_t)p->max_keypoints * p->words, p->words, (uint32_t)fper, fch, c->w_sync.as<uint32_t>(),
                         c->w_ovf.as<uint32_t>(), c->frame_flag_dev, (uint32_t)c->opt_frame_test);
      PCHK(launch_ok(c, "k_frame"));
      c->last_path = PISLAM_PATH_FUSED | PISLAM_PATH_ONE_LAUNCH;
      HIPCHK(c, hipEventRecord(c->ev[1], c->stream));   // (one launch: the stage split of last_timing is all in stage 0)
      HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
      return PISLAM_OK;
    }
  }
  const int base = batch / nsub, rem = batch % nsub;
  hipStream_t M = c->stream, X = nsub > 1 ? c->aux_stream : c->stream;
  for (int sub = 0; sub < nsub; sub++) {
    const int first = sub * base + std::min(sub, rem), n = base + (sub < rem ? 1 : 0);
    pf::FusedParams F = Fplan;
    F.batch = n;
    const uint8_t *s_pyr = pyramids + (size_t)first * stride;
    uint32_t *s_stage = c->w_stage.as<uint32_t>() + (size_t)first * F.slots_per_pyr;
    uint32_t *s_cnt = c->w_stripcnt.as<uint32_t>() + (size_t)first * S;
    uint32_t *s_sdesc = c->w_stagedesc.as<uint32_t>() + (size_t)first * sdesc_per_pyr;
    uint8_t *s_dump = dump ? dump + (size_t)first * dump_stride : nullptr;
    uint32_t *s_kp = kp + (size_t)first * p->max_keypoints;
    uint32_t *s_desc = desc + (size_t)first * p->max_keypoints * p->words;
    uint32_t *s_counts = counts + first;
    uint32_t *ovf = alias ? c->w_ovf.as<uint32_t>() + (size_t)sub * ovf_stride : nullptr;
    const dim3 grid((unsigned)(cdiv(n, 8) * F.runs_per_pyr * 8));
    unsigned long long *prof = nullptr;
    const size_t prof_n = (size_t)grid.x * 8;
    if (F.ablate & 8192) {                         // profiling hook: per-phase workgroup cycles -> stderr
      if (c->w_prof.ensure(prof_n * sizeof(unsigned long long)) != PISLAM_OK) return fail(c, PISLAM_ERR_NOMEM, "hipMalloc(prof)");
      prof = c->w_prof.as<unsigned long long>();
      HIPCHK(c, hipMemsetAsync(prof, 0, prof_n * sizeof(unsigned long long), M));
    }
    // (profiling option "repeat_strips": the strip kernel launched n times back to back inside the stage-0
    //  event bracket, so that the per-launch duration is not inflated by the command-processor latency
    //  around a single eager launch; every launch rewrites the same outputs)
    for (int rep = 0; rep < std::max(1, c->opt_repeat_strips); rep++) {
      if (rep && ovf) HIPCHK(c, hipMemsetAsync(ovf, 0, sizeof(uint32_t), M));   // the last launch's list counts
      hipLaunchKernelGGL(kern, grid, dim3(pf::NT), klds, M, F, s_pyr, stride, s_stage, s_cnt, s_dump, dump_stride, prof, ovf,
                         s_sdesc);
    }
    if (prof) {
      std::vector<unsigned long long> hv(prof_n);
      HIPCHK(c, hipMemcpyAsync(hv.data(), prof, prof_n * sizeof(unsigned long long), hipMemcpyDeviceToHost, M));
      HIPCHK(c, hipStreamSynchronize(M));
      double h[8] = {0};
      unsigned long long why[3] = {0, 0, 0};
      for (size_t i = 0; i < prof_n; i++) {
        if ((i & 7) == 5) {
          why[0] += hv[i] & 0xfffff;
          why[1] += (hv[i] >> 20) & 0xfffff;
          why[2] += hv[i] >> 40;
        } else {
          h[i & 7] += (double)hv[i];
        }
      }
      fprintf(stderr, "[pislam prof] deferred strips by reason: corner queue %llu, score queue %llu, survivor buffer %llu\n",
              why[0], why[1], why[2]);
      const double ns = (double)S * n;
      fprintf(stderr, "[pislam prof] cycles/strip: stage %.0f classify %.0f harris %.0f nms %.0f emit %.0f "
                      "(strips %.0f, carried %.0f, workgroups %u, lifetime %.0f/strip)\n",
              h[0] / ns, h[1] / ns, h[2] / ns, h[3] / ns, h[4] / ns, ns, h[6], grid.x, h[7] / ns);
    }
    PCHK(launch_ok(c, "k_fused_strips"));
    if (nsub > 1) {
      // fork: the rest of this sub-batch runs on the aux stream, under the next sub-batch's strip kernel
      HIPCHK(c, hipEventRecord(c->ev_sub[sub], M));
      HIPCHK(c, hipStreamWaitEvent(X, c->ev_sub[sub], 0));
      if (sub == nsub - 1) {
        HIPCHK(c, hipEventRecord(c->ev[1], M));   // stage 0 = every sub-batch's strip kernel
        HIPCHK(c, hipEventRecord(c->ev[2], M));   // (stage 1, the overflow passes, runs on the aux stream)
      }
    } else {
      HIPCHK(c, hipEventRecord(c->ev[1], M));     // stage 0 = the strip kernel alone
    }
    if (alias) {
      hipLaunchKernelGGL(okern, dim3((unsigned)std::max(8, c->num_cus / 2)), dim3(pf::NT), lds, X, F, s_pyr, stride, s_stage,
                         s_cnt, s_dump, dump_stride, (const uint32_t *)ovf);
      PCHK(launch_ok(c, "k_fused_overflow"));
    }
    if (nsub == 1) HIPCHK(c, hipEventRecord(c->ev[2], M));   // stage 1 = the overflow pass (normally empty)
    // the plan, lists and counts the gather runs on: the strips' own, or the units of the bucket selection pass
    pf::FusedParams G = F;
    const uint32_t *g_stage = s_stage, *g_cnt = s_cnt;
    if (sel) {
      uint32_t *u_stage = c->w_ustage.as<uint32_t>() + (size_t)first * Q.uslots_per_pyr;
      uint32_t *u_cnt = c->w_ucount.as<uint32_t>() + (size_t)first * Q.units_per_pyr;
      const size_t sel_lds = sizeof(uint32_t) * pf::SEL_WAVES * (64 + 2 * (size_t)Q.nb_max);
      hipLaunchKernelGGL(pf::k_bucket_select, dim3(cdiv(Q.units_per_pyr, pf::SEL_WAVES), n), dim3(64 * pf::SEL_WAVES), sel_lds, X, F, Q,
                         (const uint32_t *)s_stage, (const uint32_t *)s_cnt, u_stage, u_cnt, c->cur_utab);
      PCHK(launch_ok(c, "k_bucket_select"));
      G = U;
      G.batch = n;
      g_stage = u_stage;
      g_cnt = u_cnt;
    }
    if (generic_orb) {
      c->last_path |= PISLAM_PATH_GENERIC_ORB;
      hipLaunchKernelGGL(pf::k_gather, dim3(n), dim3(256), sizeof(uint32_t) * (Sg + 1), X, G, g_stage, g_cnt, s_kp,
                         (size_t)p->max_keypoints, (uint32_t)p->max_keypoints, s_counts, ovf);
      PCHK(launch_ok(c, "k_gather"));
      hipLaunchKernelGGL(pk::k_orb<0>, dim3(cdiv(p->max_keypoints, 4), 1, n), dim3(256), 0, X, s_pyr, p->vstep, stride,
                         s_kp, (size_t)p->max_keypoints, s_counts, 0u, (uint32_t)p->max_keypoints, p->words, s_desc,
                         (size_t)p->max_keypoints * p->words, (int32_t *)nullptr, (const uint8_t *)nullptr);
      PCHK(launch_ok(c, "k_orb<batch>"));
    } else {
      hipLaunchKernelGGL(gkern, dim3(nch, n), dim3(256), olds, X, G, s_pyr, stride, g_stage, g_cnt,
                         (const uint32_t *)s_sdesc, s_kp, (size_t)p->max_keypoints, (uint32_t)p->max_keypoints, s_counts,
                         s_desc, (size_t)p->max_keypoints * p->words, p->words, (uint32_t)per_max, ovf);
      PCHK(launch_ok(c, "k_gather_orb"));
    }
  }
  if (nsub > 1) {                                   // join: the call is complete, in stream order, on the context stream
    HIPCHK(c, hipEventRecord(c->ev_join, X));
    HIPCHK(c, hipStreamWaitEvent(M, c->ev_join, 0));
  }
  return PISLAM_OK;
}

}  // namespace

// Host-only: build the strip plan (and the bucket selection plan) for these parameters exactly as a batch call would —
// no device, no allocation, no launch — and check its invariants.  For the sanitizer runs of the host code (tests/
// test_sanitizers.py runs thousands of random level tables through it in a library built with -fsanitize=address,undefined)
// and for tools that want to know what a call will launch.  options: "key=value,key=value" (pislam_ctx_set_option keys).
// summary: [0] plan entries, [1] strips per pyramid, [2] runs per pyramid, [3] staging slots per pyramid, [4] run length,
// [5] LDS bytes (plain layout), [6] LDS bytes (aliased layout), [7] units of the selection pass (0: none).
PISLAM_EXPORT int pislam_debug_build_plan(const pislam_frontend_params *p, const pislam_level *lv, int batch, int num_cus,
                                          int lanes_in_flight, const char *options, uint32_t summary[8], char *err, size_t err_cap) {
  pislam_ctx c;                                       // (never touches a device: plain members only)
  auto say = [&](int rc) {
    if (err && err_cap) snprintf(err, err_cap, "%s", c.err.c_str());
    return rc;
  };
  c.num_cus = num_cus > 0 ? num_cus : 256;
  c.lanes_in_flight = lanes_in_flight > 0 ? lanes_in_flight : 1;
  if (!summary) return PISLAM_ERR_INVALID;
  memset(summary, 0, sizeof(uint32_t) * 8);
  for (const char *q = options; q && *q;) {
    const char *e = strchr(q, ','), *eq = strchr(q, '=');
    const size_t len = e ? (size_t)(e - q) : strlen(q);
    if (!eq || (size_t)(eq - q) >= len) {
      c.err = "options: key=value[,key=value...]";
      return say(PISLAM_ERR_INVALID);
    }
    const std::string key(q, eq - q);
    if (key == "own_stream") {
      c.err = "own_stream needs a device";
      return say(PISLAM_ERR_INVALID);
    }
    const int rc = pislam_ctx_set_option(&c, key.c_str(), atoi(eq + 1));
    if (rc != PISLAM_OK) return say(rc);
    q += len + (e ? 1 : 0);
  }
  int rc = check_params(&c, p, lv, batch);
  if (rc != PISLAM_OK) return say(rc);
  pf::FusedParams F;
  size_t lds = 0, lds_alias = 0;
  const int nsub = choose_sub_batches(&c, p, batch);
  if (c.opt_pipeline == 1 || !build_fused_plan(&c, p, lv, sub_max(batch, nsub), &F, &lds, &lds_alias)) {
    c.err = "no strip plan for these parameters (the staged pipeline takes the call)";
    return say(PISLAM_ERR_INVALID);
  }
  // ---- invariants the kernels rely on ----
  auto bad = [&](const char *what) {
    c.err = std::string("plan invariant violated: ") + what;
    return say(PISLAM_ERR_HIP);
  };
  if (F.nlevels < 1 || F.nlevels > pf::MAX_LEVELS) return bad("entries");
  int strips = 0, slots = 0, runs = 0;
  for (int l = 0; l < F.nlevels; l++) {
    const pf::FusedLevel &L = F.lv[l];
    if (L.strip0 != strips || L.slot0 != slots || L.run0 != runs) return bad("prefix sums");
    if (L.nstrips < 0 || (L.nstrips > 0 && (L.R < 2 || (L.R & 1)))) return bad("strip height");
    if (L.nstrips > 0) {
      if (L.col0 < 0 || L.row0 < 0 || L.col0 + L.w > p->vstep || L.row0 + L.h > p->rows) return bad("entry outside the pyramid");
      if (L.tpitch % 16 || L.pitch % 16 || L.tpitch <= 0) return bad("pitch");
      if ((L.R + 10) * L.tpitch > 150 * 1024) return bad("tile larger than the LDS");
      if (L.gfirst < 0 || L.gfirst > l || L.gn < 1 || L.gfirst + L.gn > F.nlevels) return bad("tile group");
      if (L.ex0 < p->border || L.ex1 > L.w || L.ex0 > L.ex1) return bad("owned columns");
      if (cdiv(L.nstrips, F.run_len) != L.nruns) return bad("runs");
      if ((uint64_t)L.vpr_recip * (uint64_t)(L.tpitch / 16) < (1ull << 32)) return bad("vpr_recip");
      if ((uint64_t)L.tp_recip * (uint64_t)L.tpitch < (1ull << 32)) return bad("tp_recip");
      if (L.qh < pf::QH_SHARED) return bad("corner queue");
    }
    strips += L.nstrips;
    slots += L.nstrips * (L.R / 2) * L.nbx;
    runs += L.nruns;
  }
  if (strips != F.strips_per_pyr || slots != F.slots_per_pyr || runs != F.runs_per_pyr) return bad("totals");
  if (F.order_n) {
    if (F.order_n != runs || runs > pf::MAX_ORDER) return bad("order size");
    std::vector<int> seen((size_t)runs, 0);
    for (int i = 0; i < F.order_n; i++) {
      const int e = (int)(F.order[i] >> 16), r = (int)(F.order[i] & 0xffff);
      if (e >= F.nlevels || r >= F.lv[e].nruns) return bad("order entry");
      seen[(size_t)(F.lv[e].run0 + r)]++;
    }
    for (int v : seen)
      if (v != 1) return bad("order is not a permutation of the runs");
  }
  if (lds_alias > 160 * 1024) return bad("aliased LDS size");
  summary[0] = (uint32_t)F.nlevels;
  summary[1] = (uint32_t)F.strips_per_pyr;
  summary[2] = (uint32_t)F.runs_per_pyr;
  summary[3] = (uint32_t)F.slots_per_pyr;
  summary[4] = (uint32_t)F.run_len;
  summary[5] = (uint32_t)lds;
  summary[6] = (uint32_t)lds_alias;
  if (p->log_bucket_size != 0 && F.lbs == 0 && strips > 0) {
    pf::SelectPlan Q;
    pf::FusedParams U;
    rc = build_select_plan(&c, p, F, &Q, &U);
    if (rc != PISLAM_OK) return say(rc);
    int units = 0, uslots = 0;
    for (int l = 0; l < Q.nlevels; l++) {
      if (Q.unit0[l] != units || Q.uslot0[l] != uslots) return bad("selection plan prefix sums");
      if (Q.g0[l] < 0 || Q.g0[l] + Q.gn[l] > F.nlevels) return bad("selection plan entries");
      if (Q.cap[l] / p->bucket_limit > pf::SEL_NB) return bad("more buckets than the selection pass holds");
      units += Q.nunits[l];
      uslots += Q.nunits[l] * Q.cap[l];
    }
    if (units != Q.units_per_pyr || uslots != Q.uslots_per_pyr) return bad("selection plan totals");
    summary[7] = (uint32_t)Q.units_per_pyr;
  }
  return PISLAM_OK;
}

PISLAM_EXPORT int pislam_frontend_reserve(pislam_ctx *c, const pislam_frontend_params *p,
                                          const pislam_level *lv, int batch) {
  PCHK(check_params(c, p, lv, batch));
  HIPCHK(c, hipSetDevice(c->device));
  const size_t pyr_bytes = (size_t)p->rows * p->vstep;
  bool grew = false;
  if (c->w_score.ensure(pyr_bytes * batch, &grew) != PISLAM_OK)
    return fail(c, PISLAM_ERR_NOMEM, "hipMalloc(score map)");
  const bool same_shape = !grew && c->last_batch >= batch && c->last_params.vstep == p->vstep &&
                          c->last_params.rows == p->rows && c->last_params.nlevels == p->nlevels &&
                          c->last_params.border == p->border && (int)c->last_levels.size() == p->nlevels &&
                          memcmp(c->last_levels.data(), lv, sizeof(pislam_level) * p->nlevels) == 0;
  if (!same_shape) {
    // Fast.h:42-44: `out` must start as zeros; afterwards the regions fastDetect rewrites are the
    // only ones that ever change, so the zeroing is needed once per shape.
    HIPCHK(c, hipMemsetAsync(c->w_score.p, 0, pyr_bytes * batch, c->stream));
    c->last_params = *p;
    c->last_levels.assign(lv, lv + p->nlevels);
    c->last_batch = batch;
  }
  size_t maxn = 1;
  for (int l = 0; l < p->nlevels; l++) {
    const int ny = lv[l].height - 2 * p->border, nx = lv[l].width - 2 * p->border;
    if (ny <= 0 || nx <= 0) continue;
    if (p->log_bucket_size == 0) maxn = std::max<size_t>(maxn, cdiv(ny, 2));
    else {
      const int bs = 1 << p->log_bucket_size;
      maxn = std::max<size_t>(maxn, (size_t)((nx - 1) / bs + 1) * ((ny - 1) / bs + 1));
    }
  }
  if (c->w_cnt.ensure(sizeof(uint32_t) * maxn * batch) != PISLAM_OK ||
      c->w_off.ensure(sizeof(uint32_t) * maxn * batch) != PISLAM_OK ||
      (p->log_bucket_size &&
       c->w_cellkp.ensure(sizeof(uint32_t) * maxn * batch * p->bucket_limit) != PISLAM_OK))
    return fail(c, PISLAM_ERR_NOMEM, "hipMalloc(extract scratch)");
  // fused pipeline: strip staging, strip counts, overflow list (so that the batch call itself allocates
  // nothing — it can then be captured into a hipGraph)
  if (c->opt_pipeline != 1) {
    pf::FusedParams F;
    size_t lds = 0, lds_alias = 0;
    const int nsub = choose_sub_batches(c, p, batch), submax = sub_max(batch, nsub);
    if (build_fused_plan(c, p, lv, submax, &F, &lds, &lds_alias) && F.strips_per_pyr > 0) {
      if (c->w_stage.ensure(sizeof(uint32_t) * ((size_t)F.slots_per_pyr * batch + 64)) != PISLAM_OK ||
          c->w_stripcnt.ensure(sizeof(uint32_t) * (size_t)F.strips_per_pyr * batch) != PISLAM_OK ||
          (F.orb_in_strip &&
           c->w_stagedesc.ensure(sizeof(uint32_t) * (size_t)F.strips_per_pyr * batch * pf::QS_SHARED * (size_t)p->words) != PISLAM_OK))
        return fail(c, PISLAM_ERR_NOMEM, "hipMalloc(fused staging)");
      if (c->opt_alias && submax <= 65535 && F.strips_per_pyr <= 65535)
        PCHK(prepare_ovf(c, nsub, 2 + (size_t)F.strips_per_pyr * submax));
      if (nsub > 1) PCHK(ensure_aux(c, nsub));
      if (batch <= frame_max_batch(c)) PCHK(ensure_frame_sync(c));
      if (p->log_bucket_size != 0 && F.lbs == 0) {   // the selection pass's staging (run_fused allocates nothing after this)
        pf::SelectPlan Q;
        pf::FusedParams U;
        PCHK(build_select_plan(c, p, F, &Q, &U));
        if (c->w_ustage.ensure(sizeof(uint32_t) * (size_t)Q.uslots_per_pyr * batch) != PISLAM_OK ||
            c->w_ucount.ensure(sizeof(uint32_t) * (size_t)Q.units_per_pyr * batch) != PISLAM_OK)
          return fail(c, PISLAM_ERR_NOMEM, "hipMalloc(bucket selection staging)");
        PCHK(ensure_unit_table(c, p, F, Q));
      }
    }
  }
  return PISLAM_OK;
}

PISLAM_EXPORT int pislam_orb_frontend_batch(pislam_ctx *c, const pislam_frontend_params *p,
                                            const pislam_level *lv, const uint8_t *pyramids,
                                            size_t stride, int batch, uint32_t *kp, uint32_t *desc,
                                            uint32_t *counts) {
  PCHK(check_params(c, p, lv, batch));
  PCHK(check_frame_poison(c));
  if (!pyramids || !kp || !desc || !counts) return fail(c, PISLAM_ERR_INVALID, "null pointer");
  if (stride < (size_t)p->rows * p->vstep) return fail(c, PISLAM_ERR_INVALID, "pyramid_stride too small");
  if (!is_device_ptr(pyramids) || !is_device_ptr(kp) || !is_device_ptr(desc) || !is_device_ptr(counts))
    return fail(c, PISLAM_ERR_INVALID, "the batch path takes device pointers only");
  PCHK(pislam_frontend_reserve(c, p, lv, batch));
  const size_t pyr_bytes = (size_t)p->rows * p->vstep;
  uint8_t *score = c->w_score.as<uint8_t>();
  c->last_stride = pyr_bytes;
  pf::FusedParams F;
  size_t lds = 0;
  size_t lds_alias = 0;
  const int nsub = choose_sub_batches(c, p, batch);
  bool fused = c->opt_pipeline != 1 && build_fused_plan(c, p, lv, sub_max(batch, nsub), &F, &lds, &lds_alias);
  if (c->opt_pipeline >= 2 && !fused)
    return fail(c, PISLAM_ERR_INVALID, "fused pipeline unavailable for these parameters (bucket size / LDS size)");
  c->last_pipeline = fused ? 2 : 1;
  c->last_path = fused ? PISLAM_PATH_FUSED : PISLAM_PATH_STAGED;
  HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
  if (fused && F.strips_per_pyr == 0) {            // every level is smaller than 2 x border: nothing to extract
    HIPCHK(c, hipMemsetAsync(counts, 0, sizeof(uint32_t) * batch, c->stream));
    for (int i = 1; i < 4; i++) HIPCHK(c, hipEventRecord(c->ev[i], c->stream));
    c->timing_valid = true;
    c->last_strips = 0;
    return PISLAM_OK;
  }
  if (fused) {
    PCHK(run_fused(c, p, F, lds, lds_alias, pyramids, stride, batch, nsub, kp, desc, counts));
  } else {
  HIPCHK(c, hipMemsetAsync(counts, 0, sizeof(uint32_t) * batch, c->stream));
  // The score map workspace is laid out with stride pyr_bytes; the image with `stride`.  The stage
  // kernels take one stride for both, so when they differ fall back to per-pyramid launches.
  const bool same = stride == pyr_bytes;
  for (int l = 0; l < p->nlevels; l++) {
    const size_t off = (size_t)lv[l].row0 * p->vstep + lv[l].col0;
    if (same) {
      PCHK(launch_detect(c, pyramids + off, score + off, p->vstep, pyr_bytes, batch, p->border,
                         lv[l].width, lv[l].height, p->fast_threshold));
      PCHK(launch_harris(c, pyramids + off, score + off, p->vstep, pyr_bytes, batch, p->border,
                         lv[l].width, lv[l].height, p->harris_threshold));
    } else {
      for (int b = 0; b < batch; b++) {
        PCHK(launch_detect(c, pyramids + b * stride + off, score + b * pyr_bytes + off, p->vstep, 0, 1,
                           p->border, lv[l].width, lv[l].height, p->fast_threshold));
        PCHK(launch_harris(c, pyramids + b * stride + off, score + b * pyr_bytes + off, p->vstep, 0, 1,
                           p->border, lv[l].width, lv[l].height, p->harris_threshold));
      }
    }
  }
  HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
  for (int l = 0; l < p->nlevels; l++) {
    const size_t off = (size_t)lv[l].row0 * p->vstep + lv[l].col0;
    const uint32_t add_xy = ((uint32_t)lv[l].col0 << 12) | (uint32_t)lv[l].row0;   // README.md:78
    PCHK(launch_extract(c, score + off, p->vstep, pyr_bytes, batch, p->border, p->log_bucket_size,
                        p->bucket_limit, lv[l].width, lv[l].height, kp, (size_t)p->max_keypoints,
                        (uint32_t)p->max_keypoints, add_xy, counts));
  }
  HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
  }
  if (!fused) {
    hipLaunchKernelGGL(pk::k_orb<0>, dim3(cdiv(p->max_keypoints, 4), 1, batch), dim3(256), 0, c->stream,
                       pyramids, p->vstep, stride, kp, (size_t)p->max_keypoints, counts, 0u,
                       (uint32_t)p->max_keypoints, p->words, desc, (size_t)p->max_keypoints * p->words,
                       (int32_t *)nullptr, (const uint8_t *)nullptr);
    PCHK(launch_ok(c, "k_orb<batch>"));
  }
  HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
  c->timing_valid = true;
  return PISLAM_OK;
}

PISLAM_EXPORT int pislam_frontend_get_score_map(pislam_ctx *c, int b, uint8_t *dst) {
  if (!c || !dst) return PISLAM_ERR_INVALID;
  if (b < 0 || b >= c->last_batch || !c->w_score.p || !c->last_stride)
    return fail(c, PISLAM_ERR_INVALID, "no score map for that pyramid");
  if (c->last_pipeline == 2 && !c->opt_dump_score)
    return fail(c, PISLAM_ERR_INVALID, "the fused pipeline keeps the score map in LDS (set option dump_score)");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t bytes = c->last_stride;
  HIPCHK(c, hipMemcpyAsync(dst, c->w_score.as<uint8_t>() + (size_t)b * bytes, bytes, hipMemcpyDefault,
                           c->stream));
  return sync(c);
}

PISLAM_EXPORT int pislam_frontend_last_stats(pislam_ctx *c, uint32_t stats[2]) {
  if (!c || !stats) return PISLAM_ERR_INVALID;
  stats[0] = stats[1] = 0;
  if (!c->w_ovf.p || !c->last_strips) return PISLAM_OK;      // staged pipeline / separate-tile layout: nothing deferred
  HIPCHK(c, hipSetDevice(c->device));
  PCHK(sync(c));
  PCHK(check_frame_poison(c));
  uint32_t prev[pislam_ctx::MAX_SUB] = {};
  for (int i = 0; i < c->ovf_nsub; i++)          // one list per sub-batch: [1] = strips the last step deferred
    HIPCHK(c, hipMemcpyAsync(&prev[i], c->w_ovf.as<uint32_t>() + (size_t)i * c->ovf_stride + 1, sizeof(uint32_t),
                             hipMemcpyDeviceToHost, c->stream));
  PCHK(sync(c));
  for (int i = 0; i < c->ovf_nsub; i++) stats[0] += prev[i];
  stats[1] = c->last_strips;
  return PISLAM_OK;
}

PISLAM_EXPORT unsigned pislam_frontend_last_path(const pislam_ctx *c) { return c ? c->last_path : 0u; }

PISLAM_EXPORT int pislam_frontend_last_timing(pislam_ctx *c, float *total_ms, float stage_ms[3]) {
  if (!c) return PISLAM_ERR_INVALID;
  if (!c->timing_valid) return fail(c, PISLAM_ERR_INVALID, "no batch call recorded");
  HIPCHK(c, hipEventSynchronize(c->ev[3]));
  if (total_ms) HIPCHK(c, hipEventElapsedTime(total_ms, c->ev[0], c->ev[3]));
  if (stage_ms)
    for (int i = 0; i < 3; i++) HIPCHK(c, hipEventElapsedTime(&stage_ms[i], c->ev[i], c->ev[i + 1]));
  return PISLAM_OK;
}

// ---- measurement aid: shader clock under the current load ------------------------------------------
namespace {
__global__ void k_shader_clock(unsigned long long ticks_100mhz, unsigned long long *out) {
  const unsigned long long w0 = wall_clock64(), c0 = (unsigned long long)clock64();
  unsigned long long w1 = w0;
  while (w1 - w0 < ticks_100mhz) w1 = wall_clock64();
  const unsigned long long c1 = (unsigned long long)clock64();
  if (threadIdx.x == 0) {
    out[0] = c1 - c0;
    out[1] = w1 - w0;
  }
}
}  // namespace

PISLAM_EXPORT int pislam_debug_shader_clock(pislam_ctx *c, int micros, double *ghz) {
  if (!c || !ghz) return PISLAM_ERR_INVALID;
  if (micros < 1 || micros > 100000) return fail(c, PISLAM_ERR_INVALID, "micros must be 1..100000");
  HIPCHK(c, hipSetDevice(c->device));
  if (c->w_total.ensure(2 * sizeof(unsigned long long)) != PISLAM_OK) return fail(c, PISLAM_ERR_NOMEM, "hipMalloc");
  hipLaunchKernelGGL(k_shader_clock, dim3(1), dim3(64), 0, c->stream, (unsigned long long)micros * 100ull,
                     c->w_total.as<unsigned long long>());
  PCHK(launch_ok(c, "k_shader_clock"));
  unsigned long long r[2] = {0, 0};
  HIPCHK(c, hipMemcpyAsync(r, c->w_total.p, sizeof(r), hipMemcpyDeviceToHost, c->stream));
  PCHK(sync(c));
  *ghz = r[1] ? (double)r[0] / (double)r[1] * 0.1 : 0.0;
  return PISLAM_OK;
}

// ---- descriptor matching --------------------------------------------------

namespace {

int launch_match(pislam_ctx *c, int words, const uint32_t *q, const uint32_t *qc, size_t q_stride, uint32_t nq,
                 const uint32_t *t, const uint32_t *tc, size_t t_stride, uint32_t nt, int batch, uint32_t max_q,
                 int32_t *idx, uint32_t *dist, uint32_t *dist2, size_t out_stride) {
  if (c->opt_match_mfma) {
    // matrix-core path (v_mfma_i32_32x32x32_i8): 4 waves x 32 queries per workgroup pass
    // query blocks per pair in flight: the batch API only knows the capacity (the counts live on the device), and a
    // workgroup that finds no queries still costs its launch (~30 ns each: 32 blocks per pair at batch 256 took
    // 0.27 ms against 0.09 ms with 8), so the grid aims at ~8 workgroups per CU and the workgroups loop
    const int per_pair = batch > 1 ? std::max(1, std::min(cdiv((int)max_q, pm::MF_Q), cdiv(8 * std::max(1, c->num_cus), batch))) : 65535;
    const dim3 mgrid((unsigned)std::min(cdiv((int)max_q, pm::MF_Q), per_pair), (unsigned)batch);
#define PISLAM_MATCH_MFMA(W)                                                                                  \
  hipLaunchKernelGGL(pm::k_match_mfma<W>, mgrid, dim3(64 * pm::MF_WAVES), 0, c->stream, q, qc, q_stride * W, nq, t, tc, \
                     t_stride * W, nt, (uint32_t)std::min<size_t>(q_stride, 0xffffffffu),                   \
                     (uint32_t)std::min<size_t>(t_stride, 65535), idx, dist, dist2, out_stride)
    switch (words) {
      case 1: PISLAM_MATCH_MFMA(1); break;
      case 2: PISLAM_MATCH_MFMA(2); break;
      case 4: PISLAM_MATCH_MFMA(4); break;
      case 8: PISLAM_MATCH_MFMA(8); break;
      default: return fail(c, PISLAM_ERR_INVALID, "words must be 1, 2, 4 or 8");
    }
#undef PISLAM_MATCH_MFMA
    return launch_ok(c, "k_match_mfma");
  }
  const dim3 grid((unsigned)std::min(cdiv((int)max_q, pm::QPW), batch > 1 ? pm::MAX_GRID_X : 65535), (unsigned)batch);
#define PISLAM_MATCH(W)                                                                                       \
  hipLaunchKernelGGL(pm::k_match<W>, grid, dim3(pm::QPW * pm::SPLIT), 0, c->stream, q, qc, q_stride * W, nq, t, tc, t_stride * W, \
                     nt, (uint32_t)std::min<size_t>(q_stride, 0xffffffffu), (uint32_t)std::min<size_t>(t_stride, 65535), \
                     idx, dist, dist2, out_stride)
  switch (words) {
    case 1: PISLAM_MATCH(1); break;
    case 2: PISLAM_MATCH(2); break;
    case 4: PISLAM_MATCH(4); break;
    case 8: PISLAM_MATCH(8); break;
    default: return fail(c, PISLAM_ERR_INVALID, "words must be 1, 2, 4 or 8");
  }
#undef PISLAM_MATCH
  return launch_ok(c, "k_match");
}

}  // namespace

PISLAM_EXPORT int pislam_match_hamming(pislam_ctx *c, int words, const uint32_t *query, size_t nq,
                                       const uint32_t *train, size_t nt, int32_t *idx, uint32_t *dist,
                                       uint32_t *dist2) {
  if (!c) return PISLAM_ERR_INVALID;
  if (words != 1 && words != 2 && words != 4 && words != 8) return fail(c, PISLAM_ERR_INVALID, "words must be 1, 2, 4 or 8");
  if (nt > 65535) return fail(c, PISLAM_ERR_INVALID, "at most 65535 train descriptors");
  if (nq > 0x7fffffffu) return fail(c, PISLAM_ERR_INVALID, "too many query descriptors");
  if (nq == 0) return PISLAM_OK;
  if (!query || !idx || !dist || !dist2 || (nt && !train)) return fail(c, PISLAM_ERR_INVALID, "null pointer");
  HIPCHK(c, hipSetDevice(c->device));
  Staged sq, st, si, sd, s2;
  PCHK(stage_in(c, c->s_desc, query, nq * words * sizeof(uint32_t), &sq));
  PCHK(stage_in(c, c->s_tmp, train, nt * words * sizeof(uint32_t), &st));
  PCHK(stage_in(c, c->s_pts, idx, nq * sizeof(int32_t), &si, false));
  PCHK(stage_in(c, c->s_misc, dist, nq * sizeof(uint32_t), &sd, false));
  PCHK(stage_in(c, c->s_out, dist2, nq * sizeof(uint32_t), &s2, false));
  PCHK(launch_match(c, words, (const uint32_t *)sq.dev, nullptr, nq, (uint32_t)nq, (const uint32_t *)st.dev, nullptr,
                    std::max<size_t>(nt, 1), (uint32_t)nt, 1, (uint32_t)nq, (int32_t *)si.dev, (uint32_t *)sd.dev,
                    (uint32_t *)s2.dev, nq));
  PCHK(stage_out(c, si, idx, nq * sizeof(int32_t)));
  PCHK(stage_out(c, sd, dist, nq * sizeof(uint32_t)));
  PCHK(stage_out(c, s2, dist2, nq * sizeof(uint32_t)));
  return sync(c);
}

PISLAM_EXPORT int pislam_match_hamming_batch(pislam_ctx *c, int words, const uint32_t *query,
                                             const uint32_t *qcounts, size_t q_stride, const uint32_t *train,
                                             const uint32_t *tcounts, size_t t_stride, int batch, int32_t *idx,
                                             uint32_t *dist, uint32_t *dist2) {
  if (!c) return PISLAM_ERR_INVALID;
  if (batch < 0 || batch > 65535) return fail(c, PISLAM_ERR_INVALID, "batch must be 0..65535");
  if (t_stride > 65535) return fail(c, PISLAM_ERR_INVALID, "at most 65535 train descriptors per pair");
  if (q_stride > 0x7fffffffu) return fail(c, PISLAM_ERR_INVALID, "q_stride too large");
  if (batch == 0 || q_stride == 0) return PISLAM_OK;
  if (!query || !train || !qcounts || !tcounts || !idx || !dist || !dist2) return fail(c, PISLAM_ERR_INVALID, "null pointer");
  for (const void *ptr : {(const void *)query, (const void *)train, (const void *)qcounts, (const void *)tcounts,
                          (const void *)idx, (const void *)dist, (const void *)dist2})
    if (!is_device_ptr(ptr)) return fail(c, PISLAM_ERR_INVALID, "the batch matcher takes device pointers only");
  HIPCHK(c, hipSetDevice(c->device));
  return launch_match(c, words, query, qcounts, q_stride, 0, train, tcounts, t_stride, 0, batch, (uint32_t)q_stride,
                      idx, dist, dist2, q_stride);
}

// ---- batches in flight: a pipeline of contexts behind one object --------------------------------
// One batch call at a time leaves the GPU's issue slots idle at the seams of a step (the strip kernel's tail of
// partly filled CUs, the latency-bound gather + ORB kernel, launch gaps): 0.27 ms per 256 VGA pyramids against
// 0.23 ms with three whole batches in flight.  (Overlapping INSIDE one call was built twice and measured slower
// both times — sub-batches on a second stream, strip and ORB workgroups in one grid: DESIGN.md.)  The pipeline
// object is that choreography as library API: `depth` lanes, each a context (workspace + non-blocking stream)
// of its own; batch k runs on lane k % depth, ordered after the producer of its input (an event on the caller's
// stream) and after the lane's previous batch; the caller orders its consumers with pislam_pipeline_wait.
//
// A steady stream of batches repeats its calls exactly (same buffers, same shape): a lane replays such a call from
// a hipGraph — first occurrence eager (it may size the workspace), second captured, replayed from then on
// (3 launches + 4 event records per call become one graph launch: -2.5 % per batch at depth 3).
struct LaneCall {
  pislam_frontend_params p;
  std::vector<pislam_level> lv;
  const uint8_t *pyramids;
  size_t stride;
  int batch;
  uint32_t *kp, *desc, *counts;
  hipGraphExec_t exec = nullptr;       // nullptr: seen once, not captured yet
  bool failed = false;                 // capture / instantiation failed: stay eager
  unsigned long long ws_gen = 0;       // the lane context's workspace_generation() the graph was captured against
  bool recapture = false;              // the graph was dropped because the workspace moved: next occurrence eager, then capture
  int invalidations = 0;               // times that happened: calls that keep re-laying the lane's workspace (two repeating calls
                                       // whose overflow-list layouts differ) stop being captured after 3 — each invalidation costs a
                                       // stream drain, an eager run and a recapture, more than the graph ever returns
  unsigned long long last_use = 0;
  unsigned path = 0;                   // pislam_frontend_last_path of the captured call (a replay restores it)
  int pipeline_kind = 0;
  uint32_t strips = 0;
  bool same(const pislam_frontend_params *q, const pislam_level *l, const uint8_t *py, size_t st, int b, uint32_t *k,
            uint32_t *d, uint32_t *c) const {
    return pyramids == py && stride == st && batch == b && kp == k && desc == d && counts == c &&
           memcmp(&p, q, sizeof(p)) == 0 && (int)lv.size() == q->nlevels &&
           memcmp(lv.data(), l, sizeof(pislam_level) * lv.size()) == 0;
  }
};

struct pislam_pipeline {
  int device = 0, depth = 0;
  int use_graphs = 1;                  // option "graphs"
  std::vector<std::vector<LaneCall>> calls;   // per lane, at most 4 remembered calls
  std::vector<pislam_ctx *> lane;
  std::vector<hipEvent_t> done;        // completion of the lane's last batch
  hipEvent_t in_ready = nullptr;       // producer stream -> lane stream
  unsigned long long submitted = 0;
  unsigned long long n_replayed = 0, n_captured = 0, n_capture_failed = 0, n_invalidated = 0;
  std::string err;
};

namespace {
// An exec may still be in flight on its lane: the lane's stream is drained before one is destroyed (rare events:
// an option change, an eviction, a workspace that moved).
void destroy_exec(pislam_ctx *c, LaneCall &lc) {
  if (!lc.exec) return;
  (void)hipStreamSynchronize(c->stream);
  (void)hipGraphExecDestroy(lc.exec);
  lc.exec = nullptr;
}
void drop_lane_calls(pislam_pipeline *q) {
  for (size_t li = 0; li < q->calls.size(); li++) {
    for (auto &lc : q->calls[li]) destroy_exec(q->lane[li], lc);
    q->calls[li].clear();
  }
}
}  // namespace

PISLAM_EXPORT int pislam_pipeline_destroy(pislam_pipeline *q) {
  if (!q) return PISLAM_ERR_INVALID;
  (void)hipSetDevice(q->device);
  for (auto *c : q->lane)
    if (c) (void)hipStreamSynchronize(c->stream);
  drop_lane_calls(q);
  for (auto *c : q->lane)
    if (c) (void)pislam_ctx_destroy(c);
  for (auto e : q->done)
    if (e) (void)hipEventDestroy(e);
  if (q->in_ready) (void)hipEventDestroy(q->in_ready);
  delete q;
  return PISLAM_OK;
}

PISLAM_EXPORT int pislam_pipeline_create(int device, int depth, pislam_pipeline **out) {
  if (!out || depth < 1 || depth > 8) return PISLAM_ERR_INVALID;
  *out = nullptr;
  pislam_pipeline *q = new pislam_pipeline();
  q->depth = depth;
  q->calls.resize(depth);
  for (int i = 0; i < depth; i++) {
    pislam_ctx *c = nullptr;
    int rc = pislam_ctx_create(device, &c);
    if (rc == PISLAM_OK) rc = use_own_stream(c, 2);
    if (rc == PISLAM_OK) c->lanes_in_flight = depth;
    hipEvent_t e = nullptr;
    if (rc == PISLAM_OK && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) rc = PISLAM_ERR_HIP;
    if (c) {
      q->lane.push_back(c);
      q->device = c->device;
    }
    if (e) q->done.push_back(e);
    if (rc != PISLAM_OK) {
      (void)pislam_pipeline_destroy(q);
      return rc;
    }
  }
  if (hipEventCreateWithFlags(&q->in_ready, hipEventDisableTiming) != hipSuccess) {
    (void)pislam_pipeline_destroy(q);
    return PISLAM_ERR_HIP;
  }
  *out = q;
  return PISLAM_OK;
}

PISLAM_EXPORT int pislam_pipeline_depth(const pislam_pipeline *q) { return q ? q->depth : 0; }
PISLAM_EXPORT pislam_ctx *pislam_pipeline_lane(pislam_pipeline *q, int lane) {
  return q && lane >= 0 && lane < q->depth ? q->lane[lane] : nullptr;
}
PISLAM_EXPORT void *pislam_pipeline_stream(pislam_pipeline *q, uint64_t ticket) {
  return q ? (void *)q->lane[ticket % q->depth]->stream : nullptr;
}
PISLAM_EXPORT const char *pislam_pipeline_last_error(const pislam_pipeline *q) { return q ? q->err.c_str() : "null pipeline"; }

PISLAM_EXPORT int pislam_pipeline_set_option(pislam_pipeline *q, const char *key, int value) {
  if (!q || !key) return PISLAM_ERR_INVALID;
  drop_lane_calls(q);                  // options change what a call launches: captured calls are dropped
  if (!strcmp(key, "graphs")) {        // 1 (default): repeated calls are replayed from hipGraphs; 0: always eager
    q->use_graphs = value != 0;
    return PISLAM_OK;
  }
  for (auto *c : q->lane) {
    const int rc = pislam_ctx_set_option(c, key, value);
    if (rc != PISLAM_OK) {
      q->err = c->err;
      return rc;
    }
  }
  return PISLAM_OK;
}

PISLAM_EXPORT int pislam_pipeline_reserve(pislam_pipeline *q, const pislam_frontend_params *p, const pislam_level *lv,
                                          int batch) {
  if (!q) return PISLAM_ERR_INVALID;
  for (auto *c : q->lane) {
    const int rc = pislam_frontend_reserve(c, p, lv, batch);
    if (rc != PISLAM_OK) {
      q->err = c->err;
      return rc;
    }
  }
  return PISLAM_OK;
}

PISLAM_EXPORT int pislam_pipeline_submit(pislam_pipeline *q, const pislam_frontend_params *p, const pislam_level *lv,
                                         const uint8_t *pyramids, size_t stride, int batch, uint32_t *kp, uint32_t *desc,
                                         uint32_t *counts, void *input_stream, int order_after_input, uint64_t *ticket) {
  if (!q) return PISLAM_ERR_INVALID;
  const int li = (int)(q->submitted % q->depth);
  pislam_ctx *c = q->lane[li];
  if (hipSetDevice(q->device) != hipSuccess) {
    q->err = "hipSetDevice";
    return PISLAM_ERR_HIP;
  }
  if (order_after_input) {
    // the lane's stream waits (on the device) for everything `input_stream` holds so far: the producer of `pyramids`
    hipError_t e = hipEventRecord(q->in_ready, (hipStream_t)input_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, q->in_ready, 0);
    if (e != hipSuccess) {
      q->err = std::string("pislam_pipeline_submit: ordering after the input stream: ") + hipGetErrorString(e);
      (void)hipGetLastError();
      return PISLAM_ERR_HIP;
    }
  }
  int rc = check_frame_poison(c);      // (a one-launch call of this lane timed out: reported once, graphs dropped below)
  if (rc != PISLAM_OK) {
    q->err = c->err;
    return rc;
  }
  LaneCall *hit = nullptr;
  if (q->use_graphs && p && lv && p->nlevels >= 1 && p->nlevels <= 16) {
    auto &v = q->calls[li];
    for (auto &lc : v)
      if (lc.same(p, lv, pyramids, stride, batch, kp, desc, counts)) hit = &lc;
    if (hit && hit->exec && hit->ws_gen != c->workspace_generation()) {
      // a larger call on this lane (or pislam_pipeline_reserve, or a direct call on pislam_pipeline_lane()) moved a
      // workspace buffer or re-laid the overflow lists since the capture: the graph holds stale addresses.  This
      // occurrence runs eagerly (it re-establishes the layout, as a first occurrence would), the next one captures.
      destroy_exec(c, *hit);
      hit->recapture = true;
      q->n_invalidated++;
      if (++hit->invalidations >= 3) hit->failed = true;   // (stays eager from now on)
    }
    if (hit && hit->recapture) {
      hit->recapture = false;
      hit->last_use = q->submitted;
    } else if (!hit) {                 // first occurrence: remember it, run it eagerly
      if (v.size() >= 4) {
        size_t old = 0;
        for (size_t i = 1; i < v.size(); i++)
          if (v[i].last_use < v[old].last_use) old = i;
        destroy_exec(c, v[old]);
        v.erase(v.begin() + old);
      }
      LaneCall lc;
      lc.p = *p;
      lc.lv.assign(lv, lv + p->nlevels);
      lc.pyramids = pyramids;
      lc.stride = stride;
      lc.batch = batch;
      lc.kp = kp;
      lc.desc = desc;
      lc.counts = counts;
      lc.last_use = q->submitted;
      v.push_back(lc);
    } else if (!hit->exec && !hit->failed) {          // second occurrence: capture
      hipGraph_t g = nullptr;
      hipError_t e = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
      if (e == hipSuccess) {
        rc = pislam_orb_frontend_batch(c, p, lv, pyramids, stride, batch, kp, desc, counts);
        e = hipStreamEndCapture(c->stream, &g);
        if (e == hipSuccess && rc == PISLAM_OK) e = hipGraphInstantiate(&hit->exec, g, nullptr, nullptr, 0);
        if (g) (void)hipGraphDestroy(g);
      }
      if (e != hipSuccess || rc != PISLAM_OK || !hit->exec) {
        (void)hipGetLastError();
        hit->exec = nullptr;
        hit->failed = true;            // (whatever went wrong: this call stays eager)
        rc = PISLAM_OK;
        q->n_capture_failed++;
      } else {
        hit->ws_gen = c->workspace_generation();
        hit->path = c->last_path;
        hit->pipeline_kind = c->last_pipeline;
        hit->strips = c->last_strips;
        q->n_captured++;
      }
    }
  }
  if (hit && hit->exec) {
    hit->last_use = q->submitted;
    if (hipGraphLaunch(hit->exec, c->stream) != hipSuccess) {
      q->err = "hipGraphLaunch";
      (void)hipGetLastError();
      return PISLAM_ERR_HIP;
    }
    c->timing_valid = false;           // (the timing events were recorded inside the captured call)
    c->last_path = hit->path;          // what pislam_frontend_last_path / last_stats report is the replayed call's
    c->last_pipeline = hit->pipeline_kind;
    c->last_strips = hit->strips;
    q->n_replayed++;
  } else {
    rc = pislam_orb_frontend_batch(c, p, lv, pyramids, stride, batch, kp, desc, counts);
    if (rc != PISLAM_OK) {
      q->err = c->err;
      return rc;
    }
  }
  if (hipEventRecord(q->done[li], c->stream) != hipSuccess) {
    q->err = "hipEventRecord";
    (void)hipGetLastError();
    return PISLAM_ERR_HIP;
  }
  if (ticket) *ticket = q->submitted;
  q->submitted++;
  return PISLAM_OK;
}

PISLAM_EXPORT int pislam_pipeline_wait(pislam_pipeline *q, uint64_t ticket, void *stream) {
  if (!q) return PISLAM_ERR_INVALID;
  if (ticket >= q->submitted) {
    q->err = "no such ticket";
    return PISLAM_ERR_INVALID;
  }
  // A lane's stream is in order: when the lane has taken a later batch since, its newer event covers this one.
  if (hipSetDevice(q->device) != hipSuccess || hipStreamWaitEvent((hipStream_t)stream, q->done[ticket % q->depth], 0) != hipSuccess) {
    q->err = "hipStreamWaitEvent";
    (void)hipGetLastError();
    return PISLAM_ERR_HIP;
  }
  return PISLAM_OK;
}

PISLAM_EXPORT int pislam_pipeline_stats(const pislam_pipeline *q, uint64_t stats[4]) {
  if (!q || !stats) return PISLAM_ERR_INVALID;
  stats[0] = q->submitted;
  stats[1] = q->n_replayed;
  stats[2] = q->n_captured;
  stats[3] = q->n_capture_failed;
  return PISLAM_OK;
}

PISLAM_EXPORT int pislam_pipeline_synchronize(pislam_pipeline *q) {
  if (!q) return PISLAM_ERR_INVALID;
  for (auto *c : q->lane) {
    int rc = sync(c);
    if (rc == PISLAM_OK) rc = check_frame_poison(c);
    if (rc != PISLAM_OK) {
      q->err = c->err;
      return rc;
    }
  }
  return PISLAM_OK;
}

// ---- multi-GPU: shard + count all-gather over RCCL -------------------------------
#include "pislam_dist.inc"
