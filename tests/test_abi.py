"""CPU suite: the C-ABI library loads and exports every symbol include/pislam_hip.h declares."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pislam_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pislam_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from pislam_amd import capi
    lib = capi.load()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/pislam_hip.h but not exported"
    assert set(names) == set(capi.SYMBOLS), "capi.SYMBOLS out of sync with the header"
    assert lib.pislam_abi_version() == 2


def test_no_torch_types_in_abi():
    text = open(os.path.join(ROOT, "include", "pislam_hip.h")).read()
    assert "torch" not in text.lower() and "at::" not in text


def test_brief_table_in_library_equals_compiled_reference_probe():
    from pislam_amd import capi
    ref = np.load(os.path.join(GOLDEN, "brief_table_ref.npy"))
    assert (capi.brief_table() == ref).all()


def test_product_does_not_reference_the_oracle():
    """The shipped package must never import / link the CPU checker."""
    pkg = os.path.join(ROOT, "pislam_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                t = open(os.path.join(dp, f), errors="ignore").read()
                assert "from oracle" not in t and "import oracle" not in t and "liborc" not in t, f
    for dp, _, files in os.walk(os.path.join(ROOT, "include")):
        for f in files:
            assert "oracle" not in open(os.path.join(dp, f)).read(), f


def test_dropin_headers_compile_against_reference_usage(tmp_path):
    """The README.md:59-82 pyramid loop and the demo.cpp:85-101 call sequence compile unchanged
    against include/pislam/*.h with a plain host compiler."""
    inc = os.path.join(ROOT, "include")
    if not os.path.exists(os.path.join(inc, "pislam", "Fast.h")):
        pytest.skip("drop-in headers not present yet")
    src = tmp_path / "usage.cpp"
    src.write_text(r'''
#include <vector>
#include <cstdint>
#include "pislam/Fast.h"
#include "pislam/Orb.h"
#include "pislam/Gaussian.h"
#include "pislam/Bilinear.h"
struct L { int width, height; };
static L pyramidLevels[8] = {{640,480},{533,400},{444,333},{370,278},{309,231},{257,193},{214,161},{179,134}};
static uint8_t img[2210][640];
static uint8_t out[2210][640];
int main() {
  std::vector<uint32_t> keypoints;
  std::vector<uint32_t> descriptors;
  int y = 0;
  for (int level = 0; level < 8; level += 1) {
    int oldSize = keypoints.size();
    int width =  pyramidLevels[level].width;
    int height = pyramidLevels[level].height;
    pislam::fastDetect<640, 16>(width, height, &img[y], &out[y], 20);
    pislam::fastScoreHarris<640, 16>(width, height, &img[y], 1 << 15, &out[y]);
    pislam::fastExtract<640, 16, 4, 3>(width, height, &out[y], keypoints);
    for (auto it = keypoints.begin() + oldSize; it < keypoints.end(); ++it) (*it) += y;
    y += height;
  }
  pislam::orbCompute<640, 8>(img, keypoints, descriptors);
  pislam::gaussian5x5<640>(640, 480, img, img);
  pislam::bilinear7_8<640>(640, 480, img, out);
  pislam::bilinear13_16<640>(640, 480, img, out);
  std::vector<uint32_t> p2;
  pislam::fastExtract<640, 16>(640, 480, &out[0], p2);
  uint32_t e = pislam::encodeFast(1, 2, 3);
  return (int)(pislam::decodeFastX(e) + pislam::decodeFastY(e) + pislam::decodeFastScore(e)) == 6 ? 0 : 1;
}
''')
    r = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-I", inc, str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_multi_gpu_binding_of_integration_md_compiles(tmp_path):
    """The C++ one-process-per-GPU loop INTEGRATION.md section 4 shows (pislam_dist_*: unique id, communicator,
    shard, fence, batch call, count all-gather, slowest-rank reduction) compiles against include/pislam_hip.h
    with a plain host compiler, and tools/pislam_demo.cpp (the complete program) builds."""
    src = tmp_path / "dist_usage.cpp"
    src.write_text(r'''
#include "pislam_hip.h"
int run(int rank, int world, int local_device, void *my_stream, const pislam_frontend_params &p, const pislam_level *lv,
        const uint8_t *d_pyr, size_t stride, int global_batch, uint32_t *d_kp[2], uint32_t *d_desc[2], uint32_t *d_counts[2],
        uint32_t *d_all[2], int steps) {
  uint8_t id[PISLAM_DIST_ID_BYTES] = {0};
  if (rank == 0 && pislam_dist_get_unique_id(id) != PISLAM_OK) return 1;
  pislam_ctx *ctx = nullptr;
  if (pislam_ctx_create(local_device, &ctx) != PISLAM_OK) return 2;
  pislam_ctx_set_stream(ctx, my_stream);
  if (pislam_dist_init(ctx, id, rank, world) != PISLAM_OK) return 3;
  int first = 0, count = 0;
  pislam_dist_shard(global_batch, rank, world, &first, &count);
  for (int s = 0; s < steps; s++) {
    const int o = s & 1;
    pislam_dist_fence(ctx, 2);
    pislam_orb_frontend_batch(ctx, &p, lv, d_pyr + (size_t)first * stride, stride, count, d_kp[o], d_desc[o], d_counts[o]);
    pislam_dist_allgather_counts(ctx, d_counts[o], (size_t)count, d_all[o]);
  }
  pislam_dist_synchronize(ctx);
  double t = 1.0;
  pislam_dist_allreduce_max(ctx, &t);
  const int ok = pislam_dist_rank(ctx) == rank && pislam_dist_world(ctx) == world;
  pislam_dist_finalize(ctx);
  pislam_ctx_destroy(ctx);
  return ok ? 0 : 4;
}
''')
    r = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "tools"), "pislam_demo"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_pipeline_binding_of_integration_md_compiles(tmp_path):
    """The batches-in-flight loop of INTEGRATION.md (pislam_pipeline_*: create, reserve, submit ordered after the
    producer stream, wait per ticket, one communicator for all lanes through the *_on entry points) compiles as C
    and as C++ against include/pislam_hip.h with a plain host compiler."""
    body = r'''
#include "pislam_hip.h"
int run(int device, void *my_stream, const pislam_frontend_params *p, const pislam_level *lv, const uint8_t *const *d_pyr,
        size_t stride, int batch, uint32_t *d_kp[3], uint32_t *d_desc[3], uint32_t *d_counts[3], uint32_t *d_all[3],
        pislam_ctx *comm, int nbatches) {
  pislam_pipeline *pipe = 0;
  uint64_t t[3] = {0, 0, 0};
  int k;
  if (pislam_pipeline_create(device, 3, &pipe) != PISLAM_OK) return 1;
  if (pislam_pipeline_set_option(pipe, "graphs", 1) != PISLAM_OK) return 2;
  if (pislam_pipeline_reserve(pipe, p, lv, batch) != PISLAM_OK) return 3;
  for (k = 0; k < nbatches; k++) {
    if (k >= 3) pislam_pipeline_wait(pipe, t[k % 3], my_stream);          /* the consumer of the batch three back */
    if (comm) pislam_dist_fence_on(comm, 3, pislam_pipeline_stream(pipe, (uint64_t)k));
    if (pislam_pipeline_submit(pipe, p, lv, d_pyr[k], stride, batch, d_kp[k % 3], d_desc[k % 3], d_counts[k % 3], my_stream, 1,
                               &t[k % 3]) != PISLAM_OK)
      return 4 + (pislam_pipeline_last_error(pipe) != 0);
    if (comm)
      pislam_dist_allgather_counts_on(comm, pislam_pipeline_stream(pipe, t[k % 3]), d_counts[k % 3], (size_t)batch, d_all[k % 3]);
  }
  pislam_pipeline_synchronize(pipe);
  k = pislam_pipeline_depth(pipe) == 3 && pislam_pipeline_lane(pipe, 0) != 0 && (!comm || pislam_dist_comm_count(comm) >= 0);
  pislam_pipeline_destroy(pipe);
  return k ? 0 : 9;
}
'''
    for name, cc, std in (("pipe_usage.c", "gcc", "-std=c99"), ("pipe_usage.cpp", "g++", "-std=c++11")):
        src = tmp_path / name
        src.write_text(body)
        r = subprocess.run([cc, std, "-fsyntax-only", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_compiler_resources_of_the_two_measured_kernels(tmp_path):
    """The residency DESIGN.md §5.1 / §5.2 argue with is a property of the COMPILED kernels: the product strip kernel must
    fit six waves' worth of registers (<= 80 VGPRs — five workgroups per CU are then set by its LDS alone), keep no
    private segment and at most a handful of spilled SGPRs outside its loops; k_gather_orb is pinned at an 80-VGPR
    allocation = six waves per SIMD (the step with batches in flight is measurably slower at five or seven).  Checked
    on the compiler's own resource remarks (device pass only, no GPU)."""
    import os
    import shutil
    import subprocess
    from conftest import ROOT
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    out = tmp_path / "res.txt"
    subprocess.run(["bash", os.path.join(ROOT, "tools", "kernel_resources.sh"), str(out)], check=True, timeout=600)
    rows = {}
    lines = out.read_text().splitlines()
    cols = [c.strip() for c in lines[0].split("|")]
    for ln in lines[1:]:
        f = [c.strip() for c in ln.split("|")]
        rows[f[0]] = dict(zip(cols[1:], f[1:]))
    strips = rows["void pf::k_fused_strips<true, false, true, false, false>"]
    assert int(strips["VGPRs"]) <= 80 and int(strips["ScratchSize [bytes/lane]"]) == 0, strips
    assert int(strips["VGPRs Spill"]) == 0 and int(strips["SGPRs Spill"]) <= 8, strips
    gather = rows["void pf::k_gather_orb<false>"]
    assert 73 <= int(gather["VGPRs"]) <= 80 and int(gather["ScratchSize [bytes/lane]"]) == 0, gather
    assert int(gather["VGPRs Spill"]) == 0 and int(gather["SGPRs Spill"]) == 0, gather


def test_debug_build_plan_reports_the_strip_plan_without_a_device():
    """pislam_debug_build_plan (host only: no device, no allocation): the plan of BASELINE configs[1] (59 strips per VGA pyramid
    at batch 256, 2 strips per run for a single context and whole levels for a lane of a depth-3 pipeline; 76 shorter strips and single-
    strip runs for one pyramid), the bucket selection pass's units, and the refusals (12-bit coordinates, staged pipeline)."""
    import ctypes
    from pislam_amd import capi, synth
    lib = capi.load(rebuild_if_stale=False)
    levels = synth.level_table()
    L = (capi.Level * 8)(*[capi.Level(w, h, r0, 0) for w, h, r0 in levels])

    def plan(batch, lanes=1, opts=b"", lbs=0, rows=2210):
        P = capi.FrontendParams(640, rows, 8, 16, 20, 1 << 15, lbs, 3, 8, 4096)
        out = (ctypes.c_uint32 * 8)()
        err = ctypes.create_string_buffer(200)
        rc = lib.pislam_debug_build_plan(ctypes.byref(P), L, batch, 256, lanes, opts, ctypes.byref(out), err, 200)
        return rc, list(out), err.value.decode()

    rc, s, _ = plan(256)
    assert rc == 0 and s[0] == 8 and s[1] == 59 and s[4] == 2 and s[6] <= 160 * 1024 // 5, s
    rc, s3, _ = plan(256, lanes=3)
    assert rc == 0 and s3[1] == 59 and s3[4] == 19 and s3[2] == 8, (s, s3)          # a lane: whole levels per workgroup (1.6 per slot)
    rc, s3b, _ = plan(64, lanes=3)
    assert rc == 0 and s3b[1] == 76 and s3b[4] == 9 and s3b[2] == 12, s3b                     # batch 64: the longest runs that leave 0.6 per slot
    rc, s1, _ = plan(1)
    assert rc == 0 and s1[1] == 76 and s1[4] == 1 and s1[2] == 76, s1
    rc, sb, _ = plan(256, lbs=4)
    assert rc == 0 and sb[1] == 59 and sb[7] == sum((h - 32 + 15) // 16 for _, h, _ in levels), sb   # one unit per 16-row cell row
    rc, _, msg = plan(256, opts=b"pipeline=1")
    assert rc == -1 and "staged" in msg
    rc, _, msg = plan(256, rows=5000)
    assert rc == 0 or "12 bits" in msg                                              # rows alone do not break the 12-bit rule
    Lbad = (capi.Level * 8)(*[capi.Level(w, h, r0 + 3000, 0) for w, h, r0 in levels])
    P = capi.FrontendParams(640, 6000, 8, 16, 20, 1 << 15, 0, 5, 8, 4096)
    out = (ctypes.c_uint32 * 8)()
    err = ctypes.create_string_buffer(200)
    assert lib.pislam_debug_build_plan(ctypes.byref(P), Lbad, 4, 256, 1, b"", ctypes.byref(out), err, 200) == -1
    assert b"12 bits" in err.value
