#!/usr/bin/env python3
"""Demo-equivalent CLI: what reference demo/demo.cpp:51-117 does, without libpng painting.

  python tools/pislam_demo.py pyramid.png [--buckets 4 3] [--out keypoints.npz] [--levels 640x480,533x400,...]

Loads a vertically stacked grey pyramid (PNG/PGM via PIL, or .npy), runs the reference call sequence
through the drop-in API (fastDetect -> fastScoreHarris -> fastExtract per level, then orbCompute) on the
GPU, prints per-stage wall times and the feature count (demo.cpp:113-114), optionally saves the packed
keypoints and descriptors."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

DEMO_LEVELS = [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("pyramid")
    ap.add_argument("--levels", default=None, help="WxH,WxH,... (default: the demo's 8-level VGA table)")
    ap.add_argument("--threshold", type=int, default=20)
    ap.add_argument("--harris-threshold", type=int, default=1 << 15)
    ap.add_argument("--buckets", type=int, nargs=2, default=None, metavar=("LOG_SIZE", "LIMIT"))
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    from pislam_amd import frontend as pislam

    if args.pyramid.endswith(".npy"):
        img = np.load(args.pyramid)
    else:
        from PIL import Image
        img = np.array(Image.open(args.pyramid).convert("L"))
    img = np.ascontiguousarray(img, np.uint8)
    sizes = DEMO_LEVELS if args.levels is None else [tuple(int(v) for v in t.split("x")) for t in args.levels.split(",")]
    assert img.shape[1] >= sizes[0][0] and img.shape[0] >= sum(h for _, h in sizes), "image smaller than the level table"
    out = np.zeros_like(img)
    lbs, lim = args.buckets if args.buckets else (0, 5)
    pislam.default_context()                      # device / library initialisation outside the timed part
    warm = np.zeros((64, 64), np.uint8)           # first launch loads the code object: keep it out of the timing
    pislam.fastDetect(64, 64, warm, warm.copy(), 20)
    t = {"detect": 0.0, "harris": 0.0, "extract": 0.0, "orb": 0.0}
    keypoints = []
    row = 0
    for (w, h) in sizes:
        li, lo = img[row:row + h], out[row:row + h]
        t0 = time.perf_counter(); pislam.fastDetect(w, h, li, lo, args.threshold)
        t1 = time.perf_counter(); pislam.fastScoreHarris(w, h, li, args.harris_threshold, lo)
        t2 = time.perf_counter(); kp = pislam.fastExtract(w, h, lo, logBucketSize=lbs, bucketLimit=lim)
        t3 = time.perf_counter()
        keypoints.append(kp + np.uint32(row))      # README.md:78
        t["detect"] += t1 - t0; t["harris"] += t2 - t1; t["extract"] += t3 - t2
        row += h
    keypoints = np.concatenate(keypoints).astype(np.uint32)
    t0 = time.perf_counter(); desc = pislam.orbCompute(img, keypoints); t["orb"] = time.perf_counter() - t0
    total = sum(t.values())
    print("Time: %.3f ms  (detect %.3f, harris %.3f, extract %.3f, orb %.3f; host<->device staging included)" % (
        total * 1e3, t["detect"] * 1e3, t["harris"] * 1e3, t["extract"] * 1e3, t["orb"] * 1e3))
    print(f"{len(keypoints)} features")
    if args.out:
        np.savez_compressed(args.out, keypoints=keypoints, descriptors=desc)


if __name__ == "__main__":
    main()
