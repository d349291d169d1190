"""Helper of tests/test_sanitizers.py (not a test module): runs N random front-end configurations through the library's
host-only plan builder (pislam_debug_build_plan: strip plan, launch order, bucket selection plan + their invariants).  Run
with the sanitizer build of the library (tools/asan_round.sh: PISLAM_HIP_LIB + the ASan runtime preloaded), every
out-of-bounds access, signed overflow or invalid shift of the host code on that path aborts the process.

usage: python tests/plan_fuzz.py <cases> [<first seed>]      prints "<ok> plans, <refused> refused, 0 violations" """
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pislam_amd import capi  # noqa: E402


def case(seed):
    rng = np.random.default_rng(seed)
    kind = int(rng.integers(0, 4))
    if kind == 0:                                     # a x1.2 pyramid of a random frame size, stacked
        w0, h0 = int(rng.integers(40, 2000)), int(rng.integers(40, 1200))
        nl = int(rng.integers(1, 17))
        sizes = [(max(1, int(w0 / 1.2 ** k + 0.5)), max(1, int(h0 / 1.2 ** k + 0.5))) for k in range(nl)]
        vstep = (w0 + int(rng.choice([0, 0, 1, 15, 16]))) if rng.integers(0, 2) else (w0 + 15) // 16 * 16
        levels, row = [], 0
        for w, h in sizes:
            levels.append((w, h, row, 0))
            row += h
        rows = row + int(rng.integers(0, 3))
    else:                                             # random rectangles, some side by side
        vstep = int(rng.choice([64, 192, 208, 256, 320, 640, 1280, 1920, 4096]))
        nl = int(rng.integers(1, 17))
        levels, row, col, band = [], 0, 0, 0
        for _ in range(nl):
            w, h = int(rng.integers(1, vstep + 1)), int(rng.integers(1, 400))
            if col and col + 16 + w <= vstep and rng.integers(0, 2):
                c0 = (col + 2 + 15) // 16 * 16
                levels.append((w, h, row, c0))
                col, band = c0 + w, max(band, h)
            else:
                row += band
                levels.append((w, h, row, 0))
                col, band = w, h
        rows = row + band + int(rng.integers(0, 3))
    par = dict(vstep=vstep, rows=rows, nlevels=len(levels), border=int(rng.choice([16, 16, 17, 20, 24, 40])),
               fast_threshold=int(rng.integers(0, 256)), harris_threshold=int(rng.choice([-(1 << 31), 0, 1 << 15, (1 << 31) - 1])),
               log_bucket_size=int(rng.choice([0, 0, 1, 2, 3, 4, 5, 6, 7, 8])), bucket_limit=int(rng.integers(1, 65)),
               words=int(rng.choice([1, 2, 4, 8])), max_keypoints=int(rng.choice([1, 16, 4096, 65536])))
    opts = dict(alias=int(rng.integers(0, 2)), run_len=int(rng.choice([0, 0, 1, 2, 7, 64])),
                strip_rows=int(rng.choice([0, 0, 2, 10, 16, 22, 32, 64])), tile_cols=int(rng.choice([0, 0, -1, 64, 96, 320, 448, 4096])),
                strip_rows_max=int(rng.choice([0, 0, 16, 36, 64])), bucket_select=int(rng.integers(0, 2)),
                wgs_per_cu=int(rng.choice([0, 0, 1, 3, 5, 8])), strip_px=int(rng.choice([16384, 4096, 65536])),
                run_order=int(rng.integers(0, 2)), orb_in_strip=int(rng.integers(0, 2)), sub_batches=int(rng.choice([1, 1, 0, 3, 16])),
                lds_pad=int(rng.choice([0, 0, 4096])), bucket_round_up=int(rng.integers(0, 2)))
    batch = int(rng.choice([1, 2, 3, 8, 64, 256, 4096, 70000]))
    cus = int(rng.choice([256, 256, 1, 64, 304]))
    lanes = int(rng.choice([1, 1, 3, 8]))
    return levels, par, opts, batch, cus, lanes


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    lib = capi.load(rebuild_if_stale=False)
    ok = refused = 0
    for seed in range(first, first + n):
        levels, par, opts, batch, cus, lanes = case(seed)
        P = capi.FrontendParams(*[par[k] for k in ("vstep", "rows", "nlevels", "border", "fast_threshold", "harris_threshold",
                                                   "log_bucket_size", "bucket_limit", "words", "max_keypoints")])
        L = (capi.Level * len(levels))(*[capi.Level(*t) for t in levels])
        summary = (ctypes.c_uint32 * 8)()
        err = ctypes.create_string_buffer(256)
        o = ",".join(f"{k}={v}" for k, v in opts.items()).encode()
        rc = lib.pislam_debug_build_plan(ctypes.byref(P), L, batch, cus, lanes, o, ctypes.byref(summary), err, 256)
        if rc == 0:
            ok += 1
        elif rc == -1:                                # PISLAM_ERR_INVALID: refused parameters, or the staged pipeline takes the call
            refused += 1
        else:
            print(f"seed {seed}: rc {rc}: {err.value.decode()}\n  levels {levels}\n  params {par}\n  options {opts} batch {batch} "
                  f"cus {cus} lanes {lanes}")
            return 1
    print(f"{ok} plans, {refused} refused, 0 violations ({n} cases from seed {first}; library {capi.library_path()})")
    return 0


if __name__ == "__main__":
    sys.exit(main())
