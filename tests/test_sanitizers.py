"""SURVEY §5 "sanitizers" (CPU): the oracle's whole path as a stand-alone program built with
-fsanitize=address,undefined -fno-sanitize-recover=all (oracle/asan_driver.c, `make -C oracle _asan/orc_asan`) on the
reference's demo pyramid, on adversarial levels and on buffers that end EXACTLY where the last level ends — any read or
write outside the caller's pyramid, any signed overflow / misaligned access / out-of-range shift aborts the program.
Its printed checksum must equal the un-instrumented liborc.so's results on the same input.

The product's host side: tools/asan_round.sh builds libpislam_hip with its HOST code under ASan + UBSan (device code
unchanged); here (no GPU) it runs the ABI error paths and thousands of random configurations through the host-only plan
builder (pislam_debug_build_plan: strip plan, x-tiles, launch order, bucket selection plan).  On the GPU box the suites run
against the UBSan-trap build (ROCm's ASan runtime cannot allocate device memory there): profiles/r05_ubsan_gpu.txt."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle")


@pytest.fixture(scope="module")
def asan_bin():
    subprocess.check_call(["make", "-C", ORACLE, "_asan/orc_asan"], stdout=subprocess.DEVNULL)
    return os.path.join(ORACLE, "_asan", "orc_asan")


def fnv(*arrays) -> int:
    h = 0xCBF29CE484222325
    for a in arrays:
        for b in np.ascontiguousarray(a).tobytes():
            h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def run_pyramid(asan_bin, tmp_path, img, levels4, border=16, thr=20, hthr=1 << 15, lbs=0, limit=5, words=8, cap=1 << 16):
    raw = tmp_path / "pyr.raw"
    np.ascontiguousarray(img, np.uint8).tofile(raw)
    args = [asan_bin, "pyramid", str(raw), str(img.shape[1]), str(img.shape[0]), str(border), str(thr), str(hthr), str(lbs),
            str(limit), str(words), str(cap), str(len(levels4))] + [str(int(v)) for t in levels4 for v in t]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run(args, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, f"sanitizer report or failure:\n{r.stderr[-4000:]}"
    f = dict(kv.split("=") for kv in r.stdout.split())
    return int(f["n"]), int(f["fnv"], 16)


def check(asan_bin, tmp_path, orc, img, levels4, **kw):
    cap = kw.get("cap", 1 << 16)
    n, h = run_pyramid(asan_bin, tmp_path, img, levels4, **kw)
    okw = dict(fast_threshold=kw.get("thr", 20), harris_threshold=kw.get("hthr", 1 << 15), border=kw.get("border", 16),
               log_bucket=kw.get("lbs", 0), bucket_limit=kw.get("limit", 5), words=kw.get("words", 8))
    big_kp, _, _ = orc.pyramid4(img, levels4, cap=1 << 16, **okw)
    kp, desc, _ = orc.pyramid4(img, levels4, cap=cap, **okw)
    assert n == len(big_kp) and len(kp) == min(n, cap)
    assert h == fnv(kp, desc)
    return n


def test_oracle_under_asan_ubsan_on_the_reference_demo_pyramid(asan_bin, tmp_path, orc, demo):
    lv4 = [(w, h, r0, 0) for w, h, r0 in demo["levels"]]
    assert check(asan_bin, tmp_path, orc, demo["img"], lv4) == 1754                      # SURVEY 8c pins
    assert check(asan_bin, tmp_path, orc, demo["img"], lv4, lbs=4, limit=3) == 1315
    assert check(asan_bin, tmp_path, orc, demo["img"], lv4, cap=1000, words=4) == 1754   # clipped capacity, 128-bit descriptors


def test_oracle_under_asan_ubsan_on_adversarial_levels(asan_bin, tmp_path, orc):
    """Dense noise (every queue / bucket path), binary 0/255 checkerboards and steps (saturating thresholds, extreme
    Harris sums), thresholds 0 and 255, levels packed side by side, odd sizes down to 'nothing to extract' — in a buffer
    whose last byte is the last level's last pixel."""
    rng = np.random.Generator(np.random.Philox(key=1234))
    w0, h0, vstep = 150, 97, 160
    levels4 = [(w0, h0, 0, 0), (77, 61, h0, 0), (40, 61, h0, 80), (33, 33, h0 + 61, 0), (31, 40, h0 + 61, 48)]
    rows = h0 + 61 + 40
    img = np.zeros((rows, vstep), np.uint8)
    img[:h0, :w0] = rng.integers(0, 256, (h0, w0), dtype=np.uint8)                                   # uniform noise
    yy, xx = np.mgrid[0:61, 0:77]
    img[h0:h0 + 61, :77] = (((xx // 3) + (yy // 3)) & 1) * 255                                       # checkerboard
    img[h0:h0 + 61, 80:120] = np.where(rng.integers(0, 4, (61, 40)) == 0, 255, 0)                    # binary dots
    img[h0 + 61:h0 + 94, :33] = 255                                                                  # saturated (w == 2 * border + 1)
    img[h0 + 61:, 48:79] = rng.integers(100, 140, (40, 31), dtype=np.uint8)                          # narrower than 2 * border
    img = np.ascontiguousarray(img[:rows, :])
    # the buffer ends with the last level's last row: trim the slack columns of the final row band is not possible in a
    # rectangular buffer, so a second case below puts ONE level flush against the end
    for kw in (dict(), dict(thr=0), dict(thr=255), dict(hthr=0), dict(lbs=3, limit=2), dict(lbs=5, limit=64), dict(border=18),
               dict(cap=64)):
        check(asan_bin, tmp_path, orc, img, levels4, **kw)
    flush = np.ascontiguousarray(rng.integers(0, 256, (64, 96), dtype=np.uint8))                     # vstep == width: no slack at all
    for kw in (dict(), dict(lbs=4, limit=3), dict(thr=5, hthr=-(1 << 31))):
        check(asan_bin, tmp_path, orc, flush, [(96, 64, 0, 0)], **kw)


@pytest.mark.parametrize("kind,w,h", [("gaussian", 64, 37), ("gaussian", 131, 64), ("b78", 64, 32), ("b78", 136, 40),
                                      ("b1316", 64, 32), ("b1316", 144, 48)])
def test_image_preparation_restatements_under_asan_ubsan(asan_bin, tmp_path, orc, kind, w, h):
    vstep = (w + 15) // 16 * 16
    buf = orc.fill_random(vstep, w, h)
    raw = tmp_path / "img.raw"
    buf.tofile(raw)
    r = subprocess.run([asan_bin, "prep", kind, str(raw), str(vstep), str(h), str(w), str(h)], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, f"sanitizer report or failure:\n{r.stderr[-4000:]}"
    ref = buf.copy()
    {"gaussian": orc.gaussian5x5, "b78": orc.bilinear7_8, "b1316": orc.bilinear13_16}[kind](ref, w, h)
    assert int(r.stdout.split("=")[1], 16) == fnv(ref)


def test_product_host_code_under_asan_ubsan_plan_builder_and_abi_error_paths():
    """libpislam_hip.so with its host code built -fsanitize=address,undefined (hipcc, device code unchanged), the ASan runtime
    preloaded into a child interpreter: tests/test_abi.py's error paths and 6000 random level tables / option sets through
    pislam_debug_build_plan — the code that sizes every workspace and LDS tile and fills the fixed-size plan tables
    (FusedParams::lv[24], order[160], SelectPlan[16])."""
    import glob
    rts = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    if not rts:
        pytest.skip("no ASan runtime in this ROCm")
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc: the sanitizer build of the library cannot be made here")
    # the two variants are rebuilt only when the kernel / host sources changed (their hash is kept beside them)
    from pislam_amd import build as _build
    stamp = os.path.join(ROOT, "variants", ".asan_source_hash")
    want = _build.source_hash()
    libs = [os.path.join(ROOT, "variants", n) for n in ("libpislam_hip_asan.so", "libpislam_hip_ubsan.so")]
    if not (all(os.path.exists(f) for f in libs) and os.path.exists(stamp) and open(stamp).read().strip() == want):
        b = subprocess.run(["bash", os.path.join(ROOT, "tools", "asan_round.sh"), "build"], capture_output=True, text=True)
        if b.returncode != 0 or not all(os.path.exists(f) for f in libs):
            pytest.skip(f"the sanitizer build failed here (hipcc rc {b.returncode}): {b.stderr[-500:]}")
        open(stamp, "w").write(want)
    env = dict(os.environ, LD_PRELOAD=rts[0], PISLAM_HIP_LIB=os.path.join(ROOT, "variants", "libpislam_hip_asan.so"),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([os.sys.executable, os.path.join(ROOT, "tests", "plan_fuzz.py"), "6000", "100000"], capture_output=True,
                       text=True, env=env, timeout=1200, cwd=ROOT)
    assert r.returncode == 0 and "0 violations" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert "libpislam_hip_asan.so" in r.stdout
    r = subprocess.run([os.sys.executable, "-m", "pytest", "tests/test_abi.py", "-q", "-m", "not gpu", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, env=env, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
