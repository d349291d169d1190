"""The demo-equivalent tools (reference demo/demo.cpp:51-117): tools/pislam_demo.cpp (C++ host over
the drop-in headers and the C ABI, incl. the one-process-per-GPU mode) and tools/pislam_demo.py."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, SURVEY_PINS, sha16

TOOLS = os.path.join(ROOT, "tools")


def build_tool():
    r = subprocess.run(["make", "-C", TOOLS, "pislam_demo"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return os.path.join(TOOLS, "pislam_demo")


def read_result(path):
    b = np.fromfile(path, np.uint32)
    n, m = int(b[0]), int(b[1])
    return b[2:2 + n], b[2 + n:2 + n + m]


def test_cpp_tool_builds_and_rejects_bad_input(tmp_path):
    exe = build_tool()
    bad = tmp_path / "short.raw"
    bad.write_bytes(b"\0" * 100)
    r = subprocess.run([exe, str(bad)], capture_output=True, text=True)
    assert r.returncode == 2 and "640 x 2210" in r.stderr
    assert subprocess.run([exe], capture_output=True, text=True).returncode == 1


@pytest.fixture()
def demo_files(tmp_path, demo):
    raw = tmp_path / "pyr.raw"
    demo["img"].tofile(raw)
    pgm = tmp_path / "pyr.pgm"
    with open(pgm, "wb") as f:
        f.write(b"P5\n# stacked pyramid\n640 2210\n255\n")
        f.write(demo["img"].tobytes())
    npy = tmp_path / "pyr.npy"
    np.save(npy, demo["img"])
    return raw, pgm, npy


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["raw", "pgm"])
def test_cpp_tool_dropin_sequence_reproduces_reference_pins(tmp_path, demo_files, fmt):
    """demo.cpp's call sequence through include/pislam/*.h: 1754 / 1315 features, SURVEY §8c hashes, and a
    per-stage time line like demo.cpp:113-114."""
    exe = build_tool()
    src = demo_files[0] if fmt == "raw" else demo_files[1]
    out = tmp_path / "res.bin"
    r = subprocess.run([exe, str(src), "--out", str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "1754 features" in r.stdout and re.search(r"GPU  Time: [0-9.]+ ms  \(fastDetect [0-9.]+, fastScoreHarris", r.stdout)
    kp, desc = read_result(out)
    assert len(kp) == 1754 and sha16(kp) == SURVEY_PINS["kp"] and sha16(desc) == SURVEY_PINS["desc"]
    # --paint: the demo's out.png equivalent (demo.cpp:103-111): every keypoint marked by black ticks 4-5 px away
    marked = tmp_path / "marked.pgm"
    r = subprocess.run([exe, str(src), "--paint", str(marked)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    raw = open(marked, "rb").read()
    assert raw.startswith(b"P5\n640 2210\n255\n")
    m = np.frombuffer(raw[len(b"P5\n640 2210\n255\n"):], np.uint8).reshape(2210, 640)
    exp = np.fromfile(demo_files[0], np.uint8).reshape(2210, 640).copy()
    for pt in kp:
        x, y = (int(pt) >> 12) & 0xfff, int(pt) & 0xfff
        for d in (-5, -4, 4, 5):
            exp[y + d, x] = 0
            exp[y, x + d] = 0
    assert (m == exp).all()
    r = subprocess.run([exe, str(src), "--buckets", "--out", str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "1315 features" in r.stdout, (r.stdout, r.stderr)
    assert sha16(read_result(out)[0]) == SURVEY_PINS["kp_bucket43"]


@pytest.mark.gpu
def test_cpp_tool_dropin_templates_from_several_threads(tmp_path, demo_files):
    """The reference's functions are stateless and re-entrant; the drop-in templates keep one context and one
    stream per thread (include/pislam/detail/Runtime.h), so concurrent callers neither serialise nor disturb
    each other: 4 threads, identical results, the reference's pins."""
    exe = build_tool()
    out = tmp_path / "res.bin"
    r = subprocess.run([exe, str(demo_files[0]), "--threads", "4", "--out", str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert r.stdout.count("1754 features") == 4 and "4 threads agree" in r.stdout
    kp, desc = read_result(out)
    assert sha16(kp) == SURVEY_PINS["kp"] and sha16(desc) == SURVEY_PINS["desc"]


@pytest.mark.gpu
@pytest.mark.parametrize("rccl,streams", [(False, 1), (True, 1), (False, 3), (True, 2)])
def test_cpp_tool_batch_path_and_rccl_binding(tmp_path, demo_files, rccl, streams):
    """The measured path from a C++ host: device-resident batches through ONE pislam_pipeline (--streams S lanes:
    S batches in flight, repeated calls replayed from hipGraphs inside the library), the counts through
    pislam_dist_allgather_counts_on over ONE communicator per process — with --rccl-single a real (1-rank) RCCL
    communicator created from pislam_dist_get_unique_id / pislam_dist_init (ncclCommCount reported); no Python or
    torch in the process."""
    exe = build_tool()
    out = tmp_path / "res.bin"
    cmd = [exe, str(demo_files[0]), "--batch", "6", "--steps", "7", "--streams", str(streams), "--out", str(out)] + \
        (["--rccl-single"] if rccl else [])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert f"{6 * 1754} features in 6 pyramids (1754 per pyramid)" in r.stdout
    assert ("ncclAllGather" in r.stdout) == rccl and f"{streams} in flight (pislam_pipeline)" in r.stdout
    assert f"RCCL reports {1 if rccl else 0} ranks" in r.stdout
    kp, desc = read_result(out)
    assert sha16(kp) == SURVEY_PINS["kp"] and sha16(desc) == SURVEY_PINS["desc"]


@pytest.mark.gpu
def test_cpp_tool_two_processes_two_gpus(demo_files):
    """One process per GPU from C++ (fork before HIP, id through a file, ncclCommInitRank, ncclAllGather):
    needs two GPUs — RCCL refuses two ranks on one device."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (the gpurun box has one); world 1 over RCCL is covered by --rccl-single")
    exe = build_tool()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([exe, str(demo_files[0]), "--batch", "4", "--steps", "3", "--world", "2"], capture_output=True,
                       text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert f"{8 * 1754} features in 8 pyramids" in r.stdout and "2 ranks" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("buckets", [False, True])
def test_python_cli_reproduces_reference_pins(tmp_path, demo_files, buckets):
    out = tmp_path / "kp.npz"
    cmd = [sys.executable, os.path.join(TOOLS, "pislam_demo.py"), str(demo_files[2]), "--out", str(out)]
    if buckets:
        cmd += ["--buckets", "4", "3"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    z = np.load(out)
    if buckets:
        assert "1315 features" in r.stdout and sha16(z["keypoints"]) == SURVEY_PINS["kp_bucket43"]
    else:
        assert "1754 features" in r.stdout and re.search(r"detect [0-9.]+, harris", r.stdout)
        assert sha16(z["keypoints"]) == SURVEY_PINS["kp"] and sha16(z["descriptors"]) == SURVEY_PINS["desc"]
