"""Deterministic synthetic stacked pyramids (the bench / test workload).

The reference ships no pyramid builder ("Images should be prepared by applying
a Gaussian blur and externally computing the image pyramid", reference
README.md:28-31); parity is defined on "the same stacked-pyramid input", so the
generator only has to be deterministic and to look like the reference's input
contract: a pre-blurred level 0, levels of size round(w0/1.2^k) x round(h0/1.2^k)
(the table of demo/demo.cpp:38-47), stacked vertically, left-aligned, zero
padded to `vstep`.

Level 0 = mid-grey gradient background + random rectangles / rotated boxes /
discs of random intensity + low-amplitude noise, then one 5x5 [1 4 6 4 1]/16
blur done with the rounding-halving-add tree of reference Gaussian.h
(RHADD(RHADD(b,d), RHADD(RHADD(RHADD(a,e),c),c)), reflect-101 borders).
Levels 1.. = exact-integer bilinear resample (16.16 fixed point) of level 0.

Seeds follow BASELINE.md: seed = 0x5eed0000 + pyramid index.
"""
from __future__ import annotations

import numpy as np

SEED0 = 0x5EED0000


def level_table(w0: int = 640, h0: int = 480, nlevels: int = 8, scale: float = 1.2):
    """[(width, height, row0)] for a vertically stacked pyramid (demo.cpp:38-47 rule)."""
    out, row = [], 0
    for k in range(nlevels):
        w = int(np.floor(w0 / scale ** k + 0.5))
        h = int(np.floor(h0 / scale ** k + 0.5))
        out.append((w, h, row))
        row += h
    return out


def pyramid_rows(levels) -> int:
    return int(max(t[2] + t[1] for t in levels))


def packed_level_table(w0: int = 1280, h0: int = 960, nlevels: int = 8, scale: float = 1.2, vstep: int | None = None):
    """[(width, height, row0, col0)]: like level_table but small levels share rows side by side so that
    a 1280x960 pyramid stays below the 12-bit y limit of encodeFast (Util.h:27-29).
    A level is placed right of the previous one when both fit in `vstep` with a >= 2 px zero gap
    (rounded up to a 16-byte aligned column); otherwise it starts a new row band."""
    vstep = vstep or w0
    sizes = [(int(np.floor(w0 / scale ** k + 0.5)), int(np.floor(h0 / scale ** k + 0.5))) for k in range(nlevels)]
    out, row, band_h, col = [], 0, 0, 0
    for k, (w, h) in enumerate(sizes):
        c0 = (col + 2 + 15) // 16 * 16 if col else 0
        if col and c0 + w <= vstep and k >= nlevels // 2:
            out.append((w, h, row, c0))
            band_h = max(band_h, h)
            col = c0 + w
        else:
            row += band_h
            out.append((w, h, row, 0))
            band_h, col = h, w
    return out


def _rhadd(a, b):
    return ((a.astype(np.uint16) + b.astype(np.uint16) + 1) >> 1).astype(np.uint8)


def _blur_axis(img, axis):
    n = img.shape[axis]
    idx = np.arange(-2, n + 2)
    idx = np.where(idx < 0, -idx, idx)
    idx = np.where(idx >= n, 2 * (n - 1) - idx, idx)        # reflect-101
    p = np.take(img, idx, axis=axis)
    sl = lambda o: np.take(p, np.arange(o, o + n), axis=axis)
    a, b, c, d, e = sl(0), sl(1), sl(2), sl(3), sl(4)
    return _rhadd(_rhadd(b, d), _rhadd(_rhadd(_rhadd(a, e), c), c))


def gaussian5x5(img: np.ndarray) -> np.ndarray:
    """5x5 binomial blur, RHADD tree, vertical pass then horizontal (reference
    test/GaussianTest.cpp:159-215 states the same arithmetic)."""
    return _blur_axis(_blur_axis(img, 0), 1)


def _resize_bilinear(src: np.ndarray, w: int, h: int) -> np.ndarray:
    """Exact-integer bilinear resample (pixel-centre aligned, 8-bit weights)."""
    sh, sw = src.shape
    def coords(n_out, n_in):
        # centre-aligned source coordinate in 1/256 px, clamped
        c = ((2 * np.arange(n_out, dtype=np.int64) + 1) * n_in * 256) // (2 * n_out) - 128
        c = np.clip(c, 0, (n_in - 1) * 256)
        i0 = (c >> 8).astype(np.int64)
        f = (c & 255).astype(np.uint32)
        i1 = np.minimum(i0 + 1, n_in - 1)
        return i0, i1, f
    y0, y1, fy = coords(h, sh)
    x0, x1, fx = coords(w, sw)
    s = src.astype(np.uint32)
    top = s[y0][:, x0] * (256 - fx) + s[y0][:, x1] * fx
    bot = s[y1][:, x0] * (256 - fx) + s[y1][:, x1] * fx
    out = (top * (256 - fy)[:, None] + bot * fy[:, None] + (1 << 15)) >> 16
    return out.astype(np.uint8)


def _draw_shapes(img: np.ndarray, rng: np.random.Generator, nshapes: int):
    h, w = img.shape
    kinds = rng.integers(0, 3, nshapes)
    cx = rng.integers(0, w, nshapes)
    cy = rng.integers(0, h, nshapes)
    sa = rng.integers(4, 40, nshapes)
    sb = rng.integers(4, 40, nshapes)
    ang = rng.integers(0, 180, nshapes)
    val = rng.integers(0, 256, nshapes)
    for k in range(nshapes):
        r = int(np.hypot(sa[k], sb[k])) + 1
        x0, x1 = max(0, cx[k] - r), min(w, cx[k] + r + 1)
        y0, y1 = max(0, cy[k] - r), min(h, cy[k] + r + 1)
        if x0 >= x1 or y0 >= y1:
            continue
        yy, xx = np.mgrid[y0:y1, x0:x1]
        dx, dy = xx - cx[k], yy - cy[k]
        if kinds[k] == 0:                                   # axis-aligned rectangle
            m = (np.abs(dx) <= sa[k]) & (np.abs(dy) <= sb[k])
        elif kinds[k] == 1:                                 # rotated box (integer cos/sin * 1024)
            c = int(round(np.cos(np.deg2rad(float(ang[k]))) * 1024))
            s = int(round(np.sin(np.deg2rad(float(ang[k]))) * 1024))
            u = dx * c + dy * s
            v = -dx * s + dy * c
            m = (np.abs(u) <= sa[k] * 1024) & (np.abs(v) <= sb[k] * 1024)
        else:                                               # disc
            m = dx * dx + dy * dy <= int(sa[k]) * int(sa[k])
        img[y0:y1, x0:x1][m] = val[k]


def make_level0(index: int, w0: int = 640, h0: int = 480, nshapes: int | None = None) -> np.ndarray:
    rng = np.random.Generator(np.random.Philox(key=SEED0 + index))
    if nshapes is None:
        nshapes = max(8, (w0 * h0 * 80) // (640 * 480))
    gx = rng.integers(-40, 41)
    gy = rng.integers(-40, 41)
    base = rng.integers(96, 160)
    yy, xx = np.mgrid[0:h0, 0:w0]
    img = np.clip(base + (gx * xx) // w0 + (gy * yy) // h0, 0, 255).astype(np.uint8)
    _draw_shapes(img, rng, nshapes)
    noise = rng.integers(-3, 4, size=(h0, w0))
    img = np.clip(img.astype(np.int16) + noise, 0, 255).astype(np.uint8)
    return gaussian5x5(img)


def make_pyramid(index: int, w0: int = 640, h0: int = 480, nlevels: int = 8,
                 vstep: int | None = None, levels=None, nshapes: int | None = None) -> np.ndarray:
    """uint8 [rows][vstep] stacked pyramid number `index` (seed 0x5eed0000+index)."""
    if levels is None:
        levels = level_table(w0, h0, nlevels)
    if vstep is None:
        vstep = w0
    l0 = make_level0(index, w0, h0, nshapes)
    rows = max(t[2] + t[1] for t in levels)
    out = np.zeros((rows, vstep), np.uint8)
    for t in levels:
        w, h, r0 = t[0], t[1], t[2]
        c0 = t[3] if len(t) > 3 else 0
        out[r0:r0 + h, c0:c0 + w] = l0 if (w, h) == (w0, h0) else _resize_bilinear(l0, w, h)
    return out


def make_batch(first: int, count: int, **kw) -> np.ndarray:
    """uint8 [count][rows][vstep]."""
    return np.stack([make_pyramid(first + i, **kw) for i in range(count)])


def make_many(indices, workers: int = 0, kind: str = "pyramid", **kw) -> np.ndarray:
    """The pyramids (kind="pyramid") or level-0 frames (kind="level0") numbered `indices`, generated by `workers` child
    processes (0 / 1: in this process).  The children are plain `python -m pislam_amd.synth` processes — numpy only, no
    fork of a process that holds a GPU runtime, no re-import of the parent's __main__ — that write their share to a
    temporary directory.  Same bytes as make_pyramid / make_level0 for every index."""
    indices = [int(i) for i in indices]
    one = (lambda i: make_level0(i, **kw)) if kind == "level0" else (lambda i: make_pyramid(i, **kw))
    if workers <= 1 or len(indices) < 2 * workers:
        return np.stack([one(i) for i in indices])
    import json
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    parts = [indices[k::workers] for k in range(workers)]
    with tempfile.TemporaryDirectory(prefix="pislam_synth_") as tmp:
        procs = []
        for k, part in enumerate(parts):
            spec = json.dumps({"kind": kind, "indices": part, "kw": kw, "out": os.path.join(tmp, f"part{k}.npy")})
            env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
            procs.append(subprocess.Popen([sys.executable, "-m", "pislam_amd.synth", spec], cwd=root, env=env))
        for pr in procs:
            if pr.wait() != 0:
                raise RuntimeError("a synthetic-input worker failed")
        out = [None] * len(indices)
        for k, part in enumerate(parts):
            arr = np.load(os.path.join(tmp, f"part{k}.npy"))
            for j in range(len(part)):
                out[k + j * workers] = arr[j]
    return np.stack(out)


if __name__ == "__main__":                       # a make_many worker
    import json
    import sys
    spec = json.loads(sys.argv[1])
    kw = {k: ([tuple(t) for t in v] if k == "levels" and v is not None else v) for k, v in spec["kw"].items()}
    fn = make_level0 if spec["kind"] == "level0" else make_pyramid
    np.save(spec["out"], np.stack([fn(i, **kw) for i in spec["indices"]]))
