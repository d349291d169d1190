// tools/pislam_demo.cpp — the demo-equivalent tool: what reference demo/demo.cpp:51-117
// does (load a stacked grey pyramid, fastDetect -> fastScoreHarris -> fastExtract per level, orbCompute,
// print the time and "<n> features"), as a C++ host program over the drop-in headers include/pislam/*.h
// and the C ABI include/pislam_hip.h — no libpng (raw or binary PGM in, a flat binary file out), no Python,
// no torch in the process.
//
//   pislam_demo <pyramid.raw|pyramid.pgm> [--buckets] [--out result.bin] [--paint marked.pgm]
//       the reference's call sequence through the pislam:: templates (host arrays, staged per call), with
//       the wall time of every stage — the equivalent of demo.cpp's "CPU  Time" line (demo.cpp:113-114);
//       --paint writes the pyramid with every keypoint marked by four dark ticks 4-5 pixels from it, as the
//       reference demo's out.png does (demo.cpp:103-111,118-130), as a binary PGM
//   ... --threads T
//       the same call sequence from T host threads at once (the reference's functions are re-entrant; the
//       drop-in templates keep one context + stream per thread): every thread must report the same result
//   ... --batch N [--steps K] [--streams S]
//       the measured path: N copies of the pyramid resident on the device, pislam_orb_frontend_batch
//       K times, per-stage hipEvent times (pislam_frontend_last_timing); S > 1 keeps S batches in flight: step s
//       runs on pipeline s % S (its own context, HIP stream and outputs), so one batch's gather+ORB kernel runs
//       under the next batch's strip kernel
//   ... --batch N --world W [--rccl-single]
//       one PROCESS per GPU (this program forks W ranks before touching HIP): rank 0 draws the RCCL
//       unique id (pislam_dist_get_unique_id) and hands it over through a file, every rank runs its shard of
//       the W*N pyramids and the per-pyramid counts are all-gathered (pislam_dist_allgather_counts) — the
//       C++ binding of SURVEY §8e / INTEGRATION.md §4.  --rccl-single keeps the RCCL path for W = 1.
//
// result.bin: uint32 n, uint32 n_desc_words, keypoints[n], descriptors[n_desc_words]
//
// build: make -C tools   (plain g++ with -D__HIP_PLATFORM_AMD__ against include/, libpislam_hip.so and libamdhip64)
#include <hip/hip_runtime_api.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "pislam/Fast.h"
#include "pislam/Orb.h"

namespace {

constexpr int IMG_W = 640, ROWS = 2210, NLEVELS = 8;
struct Level { int width, height; };
// the demo's level table (demo.cpp:38-47): round(640 / 1.2^k) x round(480 / 1.2^k)
const Level kLevels[NLEVELS] = {{640, 480}, {533, 400}, {444, 333}, {370, 278}, {309, 231}, {257, 193}, {214, 161}, {179, 134}};

uint8_t img[ROWS][IMG_W];

double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// raw (exactly ROWS * IMG_W bytes) or binary PGM "P5 <w> <h> <maxval<=255>"
bool load_pyramid(const char *path) {
  FILE *f = fopen(path, "rb");
  if (!f) return false;
  char magic[3] = {0, 0, 0};
  bool ok = false;
  if (fread(magic, 1, 2, f) == 2 && magic[0] == 'P' && magic[1] == '5') {
    int vals[3], got = 0, ch;
    while (got < 3 && (ch = fgetc(f)) != EOF) {
      if (ch == '#') {
        while ((ch = fgetc(f)) != EOF && ch != '\n') {}
      } else if (ch >= '0' && ch <= '9') {
        int v = ch - '0';
        while ((ch = fgetc(f)) != EOF && ch >= '0' && ch <= '9') v = 10 * v + (ch - '0');
        vals[got++] = v;               // the single whitespace after maxval has just been consumed
      }
    }
    ok = got == 3 && vals[0] == IMG_W && vals[1] == ROWS && vals[2] <= 255 && fread(img, 1, sizeof(img), f) == sizeof(img);
  } else {
    rewind(f);
    ok = fread(img, 1, sizeof(img), f) == sizeof(img);
  }
  fclose(f);
  return ok;
}

// the demo's marker (demo.cpp:118-130): black ticks at distance 4 and 5 above, below, left and right
void paint_point(int x, int y) {
  const int d[4] = {-5, -4, 4, 5};
  for (int k = 0; k < 4; k++) {
    if (y + d[k] >= 0 && y + d[k] < ROWS) img[y + d[k]][x] = 0;
    if (x + d[k] >= 0 && x + d[k] < IMG_W) img[y][x + d[k]] = 0;
  }
}
bool write_pgm(const char *path) {
  FILE *o = fopen(path, "wb");
  if (!o) return false;
  fprintf(o, "P5\n%d %d\n255\n", IMG_W, ROWS);
  const bool ok = fwrite(img, 1, sizeof(img), o) == sizeof(img);
  fclose(o);
  return ok;
}

void write_result(const char *path, const std::vector<uint32_t> &kp, const std::vector<uint32_t> &desc) {
  FILE *o = fopen(path, "wb");
  if (!o) return;
  const uint32_t n = (uint32_t)kp.size(), m = (uint32_t)desc.size();
  fwrite(&n, 4, 1, o);
  fwrite(&m, 4, 1, o);
  fwrite(kp.data(), 4, n, o);
  fwrite(desc.data(), 4, m, o);
  fclose(o);
}

#define HIP_OK(call)                                                                     \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess) {                                                              \
      fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_));                        \
      return 10;                                                                         \
    }                                                                                    \
  } while (0)
#define PISLAM_OK_(ctx, call)                                                            \
  do {                                                                                   \
    int r_ = (call);                                                                     \
    if (r_ != PISLAM_OK) {                                                               \
      fprintf(stderr, "%s -> %d (%s)\n", #call, r_, pislam_last_error(ctx));              \
      return 11;                                                                         \
    }                                                                                    \
  } while (0)

// the reference's call sequence through the drop-in templates, timed per stage
int run_dropin(bool buckets, const char *out_path, std::vector<uint32_t> *points_out = nullptr,
               const char *paint_path = nullptr) {
  std::vector<uint32_t> points, descriptors;
  std::vector<uint8_t> out_buf((size_t)ROWS * IMG_W, 0);     // the score map `out`, zero-initialised (Fast.h:42-44)
  uint8_t(*out)[IMG_W] = (uint8_t(*)[IMG_W])out_buf.data();
  pislam::detail::runtime();                          // device / library start-up outside the timed part
  {
    static uint8_t warm_in[64][IMG_W], warm_out[64][IMG_W];   // the first launch loads the code object
    pislam::fastDetect<IMG_W, 16>(64, 64, warm_in, warm_out, 20);
  }
  double t_detect = 0, t_harris = 0, t_extract = 0, t_orb = 0;
  const double begin = now_ms();
  uint32_t pyramidRow = 0;
  for (int l = 0; l < NLEVELS; l++) {
    const int w = kLevels[l].width, h = kLevels[l].height;
    uint8_t(*imgPtr)[IMG_W] = &img[pyramidRow];
    uint8_t(*outPtr)[IMG_W] = &out[pyramidRow];
    const double t0 = now_ms();
    pislam::fastDetect<IMG_W, 16>(w, h, imgPtr, outPtr, 20);
    const double t1 = now_ms();
    pislam::fastScoreHarris<IMG_W, 16>(w, h, imgPtr, 1 << 15, outPtr);
    const double t2 = now_ms();
    const size_t oldSize = points.size();
    if (buckets)
      pislam::fastExtract<IMG_W, 16, 4, 3>(w, h, outPtr, points);     // README.md:45,76
    else
      pislam::fastExtract<IMG_W, 16>(w, h, outPtr, points);
    const double t3 = now_ms();
    for (size_t i = oldSize; i < points.size(); i++)                  // demo.cpp:92-97 / README.md:78
      points[i] = pislam::encodeFast(pislam::decodeFastScore(points[i]), pislam::decodeFastX(points[i]),
                                     pislam::decodeFastY(points[i]) + pyramidRow);
    t_detect += t1 - t0;
    t_harris += t2 - t1;
    t_extract += t3 - t2;
    pyramidRow += (uint32_t)h;
  }
  const double t4 = now_ms();
  pislam::orbCompute<IMG_W, 8>(img, points, descriptors);
  const double end = now_ms();
  t_orb = end - t4;
  printf("GPU  Time: %.3f ms  (fastDetect %.3f, fastScoreHarris %.3f, fastExtract %.3f, orbCompute %.3f; "
         "host<->device staging of every call included)\n", end - begin, t_detect, t_harris, t_extract, t_orb);
  printf("%zu features\n", points.size());
  if (out_path) write_result(out_path, points, descriptors);
  if (points_out) {
    *points_out = points;
    points_out->insert(points_out->end(), descriptors.begin(), descriptors.end());
  }
  if (paint_path) {                                   // after everything that reads the image
    for (uint32_t pt : points) paint_point((int)pislam::decodeFastX(pt), (int)pislam::decodeFastY(pt));
    if (!write_pgm(paint_path)) return 16;
  }
  return 0;
}

// the measured path from C++: device-resident batch, optional shard over `world` processes, `streams` batches in
// flight through ONE pislam_pipeline (lanes of contexts inside the library; repeated calls replayed from hipGraphs)
// and ONE communicator per process (a context of its own; the all-gathers are ordered after / fence the lanes'
// streams with pislam_dist_allgather_counts_on / pislam_dist_fence_on)
struct LibOpt {
  std::string key;
  int value = 0;
};
LibOpt g_opts[8];
int g_nopts = 0;

struct OutSet {
  uint32_t *d_kp = nullptr, *d_desc = nullptr, *d_counts = nullptr, *d_all = nullptr;
};

int run_batch(int batch, int steps, bool buckets, const char *out_path, int rank, int world, const char *id_file,
              bool rccl_single, int streams) {
  int ndev = 0;
  HIP_OK(hipGetDeviceCount(&ndev));
  if (ndev < 1) {
    fprintf(stderr, "no HIP device\n");
    return 10;
  }
  if (world > ndev && !(world == 1)) {
    fprintf(stderr, "--world %d needs %d GPUs (one process per GPU), %d visible\n", world, world, ndev);
    return 12;
  }
  const int device = rank % ndev;
  HIP_OK(hipSetDevice(device));
  pislam_pipeline *pipe = nullptr;
  if (pislam_pipeline_create(device, streams, &pipe) != PISLAM_OK) return 11;
  for (int i = 0; i < g_nopts; i++)
    if (pislam_pipeline_set_option(pipe, g_opts[i].key.c_str(), g_opts[i].value) != PISLAM_OK) {
      fprintf(stderr, "--opt %s=%d: %s\n", g_opts[i].key.c_str(), g_opts[i].value, pislam_pipeline_last_error(pipe));
      return 11;
    }
  pislam_ctx *comm = nullptr;                            // holds the communicator and the collective stream
  if (pislam_ctx_create(device, &comm) != PISLAM_OK) return 11;
  // ---- one process per GPU: ONE communicator from a unique id handed over through a file ----
  const bool need_id = world > 1 || rccl_single;
  uint8_t id[PISLAM_DIST_ID_BYTES];
  memset(id, 0, sizeof(id));
  if (need_id) {
    const std::string path = std::string(id_file) + ".0";
    if (rank == 0) {
      if (pislam_dist_get_unique_id(id) != PISLAM_OK) {
        fprintf(stderr, "pislam_dist_get_unique_id failed (RCCL not loadable?)\n");
        return 13;
      }
      const std::string tmp = path + ".tmp";
      FILE *f = fopen(tmp.c_str(), "wb");
      if (!f || fwrite(id, 1, sizeof(id), f) != sizeof(id)) return 13;
      fclose(f);
      rename(tmp.c_str(), path.c_str());               // atomic: readers never see a partial id
    } else {
      FILE *f = nullptr;
      for (int tries = 0; tries < 6000 && !(f = fopen(path.c_str(), "rb")); tries++) usleep(10000);
      if (!f || fread(id, 1, sizeof(id), f) != sizeof(id)) {
        fprintf(stderr, "rank %d: no unique id at %s\n", rank, path.c_str());
        return 13;
      }
      fclose(f);
    }
    if (rccl_single) PISLAM_OK_(comm, pislam_ctx_set_option(comm, "dist_rccl_single", 1));
  }
  PISLAM_OK_(comm, pislam_dist_init(comm, need_id ? id : nullptr, rank, world));
  const int rccl_ranks = pislam_dist_comm_count(comm);   // what RCCL itself reports (0: no communicator)

  // ---- this rank's shard of the world * batch pyramids (all copies of the one input here) ----
  int first = 0, count = 0;
  pislam_dist_shard(world * batch, rank, world, &first, &count);
  pislam_level lv[NLEVELS];
  int row = 0;
  for (int l = 0; l < NLEVELS; l++) {
    lv[l] = {kLevels[l].width, kLevels[l].height, row, 0};
    row += kLevels[l].height;
  }
  pislam_frontend_params P = {IMG_W, ROWS, NLEVELS, 16, 20, 1 << 15, buckets ? 4 : 0, buckets ? 3 : 5, 8, 4096};
  const size_t pyr_bytes = (size_t)ROWS * IMG_W;
  // every lane reads its OWN copy of the batch (the photo x count): batches in flight never share input cache lines,
  // as a stream of different frames would not either
  std::vector<uint8_t *> d_in((size_t)streams, nullptr);
  for (uint8_t *&d : d_in) {
    HIP_OK(hipMalloc(&d, pyr_bytes * count));
    for (int b = 0; b < count; b++) HIP_OK(hipMemcpy(d + b * pyr_bytes, img, pyr_bytes, hipMemcpyHostToDevice));
  }
  std::vector<OutSet> sets((size_t)streams);            // one output set per lane
  for (OutSet &o : sets) {
    HIP_OK(hipMalloc(&o.d_kp, sizeof(uint32_t) * (size_t)P.max_keypoints * count));
    HIP_OK(hipMalloc(&o.d_desc, sizeof(uint32_t) * (size_t)P.max_keypoints * P.words * count));
    HIP_OK(hipMalloc(&o.d_counts, sizeof(uint32_t) * count));
    HIP_OK(hipMalloc(&o.d_all, sizeof(uint32_t) * (size_t)count * world));
  }
  if (pislam_pipeline_reserve(pipe, &P, lv, count) != PISLAM_OK) {
    fprintf(stderr, "pislam_pipeline_reserve: %s\n", pislam_pipeline_last_error(pipe));
    return 15;
  }
  auto submit = [&](int s) -> int {
    OutSet &o = sets[(size_t)(s % streams)];
    uint64_t t = 0;
    // the lane's stream is in order, so its previous batch is behind it; the all-gather that READ this output set
    // (`streams` exchanges ago) must be done before the batch overwrites it
    void *lane_stream = pislam_pipeline_stream(pipe, (uint64_t)s);       // (lane s % streams)
    const bool exchange = world > 1 || rccl_single || batch > 2;        // (a frame at a time on one GPU: nothing to exchange)
    if (exchange && pislam_dist_fence_on(comm, streams, lane_stream) != PISLAM_OK) return 1;
    if (pislam_pipeline_submit(pipe, &P, lv, d_in[(size_t)(s % streams)], pyr_bytes, count, o.d_kp, o.d_desc, o.d_counts, nullptr, 0, &t) != PISLAM_OK) {
      fprintf(stderr, "pislam_pipeline_submit: %s\n", pislam_pipeline_last_error(pipe));
      return 1;
    }
    if (!exchange) return 0;
    return pislam_dist_allgather_counts_on(comm, pislam_pipeline_stream(pipe, t), o.d_counts, (size_t)count, o.d_all) != PISLAM_OK;
  };
  for (int s = 0; s < 3 * streams; s++)                  // warm-up: eager, capture, first replay on every lane
    if (submit(s)) return 15;
  if (pislam_pipeline_synchronize(pipe) != PISLAM_OK) return 15;
  PISLAM_OK_(comm, pislam_dist_synchronize(comm));

  double barrier = 0;
  PISLAM_OK_(comm, pislam_dist_allreduce_max(comm, &barrier));      // all ranks start together
  const double t0 = now_ms();
  for (int s = 0; s < steps; s++)
    if (submit(3 * streams + s)) return 15;
  if (pislam_pipeline_synchronize(pipe) != PISLAM_OK) return 15;
  PISLAM_OK_(comm, pislam_dist_synchronize(comm));
  double dt = now_ms() - t0;
  PISLAM_OK_(comm, pislam_dist_allreduce_max(comm, &dt));           // the slowest rank

  std::vector<uint32_t> all((size_t)count * world);
  const bool exchanged = world > 1 || rccl_single || batch > 2;
  HIP_OK(hipMemcpy(all.data(), exchanged ? sets[0].d_all : sets[0].d_counts, all.size() * 4, hipMemcpyDeviceToHost));
  unsigned long long total = 0;
  for (uint32_t c : all) total += c < (uint32_t)P.max_keypoints ? c : (uint32_t)P.max_keypoints;
  int bad = 0;
  for (size_t k = 1; k < sets.size(); k++) {            // every lane gathered the same counts
    std::vector<uint32_t> other(all.size());
    HIP_OK(hipMemcpy(other.data(), exchanged ? sets[k].d_all : sets[k].d_counts, other.size() * 4, hipMemcpyDeviceToHost));
    bad += other != all;
  }
  if (rank == 0) {
    printf("GPU  Time: %.3f ms per batch of %d x %d pyramids, %d in flight (pislam_pipeline); %d ranks, count all-gather: %s, "
           "RCCL reports %d ranks\n",
           dt / steps, world, count, streams, world,
           (world > 1 || rccl_single) ? "ncclAllGather via pislam_dist_allgather_counts_on, one communicator" : "single GPU",
           rccl_ranks);
    printf("%llu features in %d pyramids (%u per pyramid), %.3e features/s\n", total, world * count, all[0],
           (double)total * steps / (dt * 1e-3));
    if (out_path) {                                      // pyramid 0 of rank 0
      const OutSet &o = sets[0];
      const uint32_t n = all[0] < (uint32_t)P.max_keypoints ? all[0] : (uint32_t)P.max_keypoints;
      std::vector<uint32_t> kp(n), desc((size_t)n * P.words);
      HIP_OK(hipMemcpy(kp.data(), o.d_kp, n * 4, hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(desc.data(), o.d_desc, desc.size() * 4, hipMemcpyDeviceToHost));
      write_result(out_path, kp, desc);
    }
  }
  // every pyramid is the same image here, so every gathered count must be equal — on every rank
  for (uint32_t c : all) bad += c != all[0];
  pislam_dist_finalize(comm);
  pislam_ctx_destroy(comm);
  pislam_pipeline_destroy(pipe);
  for (OutSet &o : sets) {
    (void)hipFree(o.d_kp); (void)hipFree(o.d_desc); (void)hipFree(o.d_counts); (void)hipFree(o.d_all);
  }
  for (uint8_t *d : d_in) (void)hipFree(d);
  return bad ? 14 : 0;
}

}  // namespace

int main(int argc, char **argv) {
  if (argc < 2) {
    fprintf(stderr, "Usage: %s pyramid.raw|pyramid.pgm [--buckets] [--out result.bin] [--paint marked.pgm] [--threads T] [--batch N [--steps K] "
                    "[--streams S] [--world W] [--rccl-single] [--opt key=value ...]]\n", argv[0]);
    return 1;
  }
  bool buckets = false, rccl_single = false;
  const char *out_path = nullptr, *paint_path = nullptr;
  int batch = 0, steps = 10, world = 1, threads = 1, streams = 3;
  for (int i = 2; i < argc; i++) {
    if (!strcmp(argv[i], "--buckets")) buckets = true;
    else if (!strcmp(argv[i], "--rccl-single")) rccl_single = true;
    else if (!strcmp(argv[i], "--out") && i + 1 < argc) out_path = argv[++i];
    else if (!strcmp(argv[i], "--paint") && i + 1 < argc) paint_path = argv[++i];
    else if (!strcmp(argv[i], "--batch") && i + 1 < argc) batch = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--steps") && i + 1 < argc) steps = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--world") && i + 1 < argc) world = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--threads") && i + 1 < argc) threads = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--streams") && i + 1 < argc) streams = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--opt") && i + 1 < argc && g_nopts < 8 && strchr(argv[i + 1], '=')) {   // library option key=value (batch path)
      const char *kv = argv[++i];
      g_opts[g_nopts].key.assign(kv, strchr(kv, '=') - kv);
      g_opts[g_nopts++].value = atoi(strchr(kv, '=') + 1);
    }
    else {
      fprintf(stderr, "unknown argument %s\n", argv[i]);
      return 1;
    }
  }
  if (!load_pyramid(argv[1])) {
    fprintf(stderr, "%s: expected a raw or binary-PGM grey image of %d x %d (the demo's stacked 8-level pyramid)\n",
            argv[1], IMG_W, ROWS);
    return 2;
  }
  if (batch <= 0 && threads > 1) {
    std::vector<std::vector<uint32_t>> res((size_t)threads);
    std::vector<int> rc((size_t)threads, 0);
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++)
      pool.emplace_back([&, t] { rc[t] = run_dropin(buckets, t == 0 ? out_path : nullptr, &res[t]); });
    for (auto &th : pool) th.join();
    for (int t = 0; t < threads; t++)
      if (rc[t] != 0 || res[t] != res[0]) {
        fprintf(stderr, "thread %d disagrees with thread 0\n", t);
        return 15;
      }
    printf("%d threads agree\n", threads);
    return 0;
  }
  if (batch <= 0) return run_dropin(buckets, out_path, nullptr, paint_path);
  if (world < 1 || steps < 1 || streams < 1 || streams > 8) return 1;
  // one process per GPU: fork the ranks BEFORE the first HIP call (a forked HIP runtime is unusable)
  char id_file[256];
  snprintf(id_file, sizeof(id_file), "/tmp/pislam_demo_id_%d", (int)getpid());
  if (world == 1) {
    const int rc = run_batch(batch, steps, buckets, out_path, 0, 1, id_file, rccl_single, streams);
    for (int k = 0; k < streams; k++) unlink((std::string(id_file) + "." + std::to_string(k)).c_str());
    return rc;
  }
  std::vector<pid_t> kids;
  for (int r = 0; r < world; r++) {
    const pid_t pid = fork();
    if (pid < 0) return 20;
    if (pid == 0) _exit(run_batch(batch, steps, buckets, out_path, r, world, id_file, false, streams));
    kids.push_back(pid);
  }
  int rc = 0;
  for (pid_t k : kids) {
    int st = 0;
    waitpid(k, &st, 0);
    if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = WIFEXITED(st) ? WEXITSTATUS(st) : 21;
  }
  for (int k = 0; k < streams; k++) unlink((std::string(id_file) + "." + std::to_string(k)).c_str());
  return rc;
}
