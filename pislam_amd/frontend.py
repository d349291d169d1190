"""Host-side mirror of the reference's public interface for the ORB hot path.

Same names, argument order and semantics as the header-only templates of
0xfaded/pislam (include/Fast.h, Harris.h, Orb.h, Brief.h, Util.h); the template
parameters (vstep, border, logBucketSize, bucketLimit, words) become keyword
arguments, `vstep` is taken from the array's row stride.  Every function forwards
to the C ABI of libpislam_hip.so — nothing is computed in Python and nothing
falls back to the CPU.

Arrays may be numpy (host; staged through the device by the library) or torch
CUDA/HIP tensors (used in place).
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import capi
from .capi import Context, FrontendParams, Level, ptr

_default_ctx: Context | None = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context()
    return _default_ctx


# ---- Util.h:27-45 ----------------------------------------------------------
def encodeFast(score, x, y):
    return ((np.uint32(score) << np.uint32(24)) | (np.uint32(x) << np.uint32(12)) | np.uint32(y)).astype(np.uint32) \
        if isinstance(score, np.ndarray) else ((int(score) << 24) | (int(x) << 12) | int(y)) & 0xFFFFFFFF


def rencodeFastScore(score, encoded):
    return ((int(score) << 24) | (int(encoded) & 0xFFFFFF)) & 0xFFFFFFFF


def decodeFastX(encoded):
    return (encoded >> 12) & 0xFFF


def decodeFastY(encoded):
    return encoded & 0xFFF


def decodeFastScore(encoded):
    return encoded >> 24


def _vstep(a) -> int:
    if a.ndim != 2:
        raise ValueError("image / score map must be 2-D [rows][vstep]")
    return int(a.shape[1])


# ---- Fast.h:54 ---------------------------------------------------------------
def fastDetect(width, height, img, out, threshold, *, border=16, ctx: Context | None = None):
    """pislam::fastDetect<vstep,border>(width, height, img, out, threshold)."""
    ctx = ctx or default_context()
    vstep = _vstep(img)
    if isinstance(img, np.ndarray) and width > 2 * border and height > 2 * border:
        # flat addressing of the over-classified columns (see pislam_hip.h): never read past the array
        need = (height - border + 2) * vstep + border + 16 * (-(-(width - 2 * border) // 16)) + 3
        if need > img.size:
            pad = np.zeros(need, np.uint8)
            pad[:img.size] = img.reshape(-1)
            img = pad
    ctx.check(ctx.lib.pislam_fast_detect(ctx.h, vstep, border, width, height, ptr(img), ptr(out),
                                         threshold), "pislam_fast_detect")


# ---- Fast.h:166 --------------------------------------------------------------
def fastScoreHarris(width, height, img, threshold, out, *, border=16, ctx: Context | None = None):
    """pislam::fastScoreHarris<vstep,border>(width, height, img, threshold, out)."""
    ctx = ctx or default_context()
    ctx.check(ctx.lib.pislam_fast_score_harris(ctx.h, _vstep(img), border, width, height, ptr(img),
                                               threshold, ptr(out)), "pislam_fast_score_harris")


# ---- Fast.h:196 --------------------------------------------------------------
def fastExtract(width, height, out, results: list | None = None, *, border=16, logBucketSize=0,
                bucketLimit=5, ctx: Context | None = None) -> np.ndarray:
    """pislam::fastExtract<vstep,border,logBucketSize,bucketLimit>(width, height, out, results).

    Appends to `results` (a Python list, like the reference's std::vector&) when given and
    returns the keypoints of this call as a uint32 array."""
    ctx = ctx or default_context()
    cap = max(16, ((width + 1) // 2) * ((height + 1) // 2))
    buf = np.zeros(cap, np.uint32)
    n = ctypes.c_size_t(0)
    ctx.check(ctx.lib.pislam_fast_extract(ctx.h, _vstep(out), border, logBucketSize, bucketLimit, width,
                                          height, ptr(out), ptr(buf), cap, ctypes.byref(n)),
              "pislam_fast_extract")
    kp = buf[:n.value].copy()
    if results is not None:
        results.extend(int(v) for v in kp)
    return kp


# ---- Harris.h:80 ---------------------------------------------------------------
def harrisScoreSobel(img, x, y, threshold, *, ctx: Context | None = None) -> int:
    """pislam::harrisScoreSobel<vstep>(img, x, y, threshold)."""
    return int(harrisScorePoints(img, np.array([(int(x) << 12) | int(y)], np.uint32), threshold, ctx=ctx)[0])


def harrisScorePoints(img, points, threshold, *, ctx: Context | None = None) -> np.ndarray:
    ctx = ctx or default_context()
    points = np.ascontiguousarray(points, np.uint32)
    scores = np.zeros(len(points), np.uint8)
    ctx.check(ctx.lib.pislam_harris_score_points(ctx.h, _vstep(img), ptr(img), ptr(points), len(points),
                                                 threshold, ptr(scores)), "pislam_harris_score_points")
    return scores


# ---- Orb.h:80 ------------------------------------------------------------------
def orbCentroids(img, points, *, ctx: Context | None = None) -> np.ndarray:
    """pislam::orbCentroids<vstep>(img, points): int32, groups [x0 x1 x2 x3 y0 y1 y2 y3]."""
    ctx = ctx or default_context()
    points = np.ascontiguousarray(points, np.uint32)
    n8 = ctx.lib.pislam_centroids_size(len(points))
    cen = np.zeros(n8, np.int32)
    ctx.check(ctx.lib.pislam_orb_centroids(ctx.h, _vstep(img), ptr(img), ptr(points), len(points),
                                           ptr(cen)), "pislam_orb_centroids")
    return cen


# ---- Orb.h:310 -----------------------------------------------------------------
def atan2(xys, *, ctx: Context | None = None) -> np.ndarray:
    """pislam::atan2(const std::vector<int32_t>&): uint8 angle bins, padding slots included."""
    ctx = ctx or default_context()
    xys = np.ascontiguousarray(xys, np.int32)
    ang = np.zeros(len(xys) // 2, np.uint8)
    ctx.check(ctx.lib.pislam_orb_angles(ctx.h, ptr(xys), len(xys), ptr(ang)), "pislam_orb_angles")
    return ang


# ---- Brief.h:637 ---------------------------------------------------------------
def briefDescribe(img, x, y, rot, *, words=8, ctx: Context | None = None) -> np.ndarray:
    """pislam::briefDescribe<vstep,words>(img, x, y, rot, descriptor)."""
    pts = np.array([(int(x) << 12) | int(y)], np.uint32)
    return briefDescribePoints(img, pts, np.array([rot], np.uint8), words=words, ctx=ctx)[0]


def briefDescribePoints(img, points, rots, *, words=8, ctx: Context | None = None) -> np.ndarray:
    ctx = ctx or default_context()
    points = np.ascontiguousarray(points, np.uint32)
    rots = np.ascontiguousarray(rots, np.uint8)
    desc = np.zeros((len(points), words), np.uint32)
    ctx.check(ctx.lib.pislam_brief_describe(ctx.h, _vstep(img), words, ptr(img), ptr(points), ptr(rots),
                                            len(points), ptr(desc)), "pislam_brief_describe")
    return desc


# ---- Orb.h:396 -----------------------------------------------------------------
def orbCompute(img, points, descriptors: list | None = None, *, words=8,
               ctx: Context | None = None) -> np.ndarray:
    """pislam::orbCompute<vstep,words>(img, points, descriptors); returns uint32 [n][words] and
    appends the flattened words to `descriptors` when given."""
    ctx = ctx or default_context()
    points = np.ascontiguousarray(points, np.uint32)
    desc = np.zeros((len(points), words), np.uint32)
    ctx.check(ctx.lib.pislam_orb_compute(ctx.h, _vstep(img), words, ptr(img), ptr(points), len(points),
                                         ptr(desc)), "pislam_orb_compute")
    if descriptors is not None:
        descriptors.extend(int(v) for v in desc.reshape(-1))
    return desc


# ---- descriptor matching (no reference counterpart) -----------------
def matchHamming(query, train, *, ctx: Context | None = None):
    """Brute-force Hamming matching of uint32 [n][words] descriptors (numpy or torch): returns
    (idx int32 [nq], dist uint32 [nq], dist2 uint32 [nq]) — nearest train index (ties: smallest
    index, -1 without train descriptors), its distance, and the best distance among the others."""
    ctx = ctx or default_context()
    query = np.ascontiguousarray(query, np.uint32)
    train = np.ascontiguousarray(train, np.uint32)
    words = query.shape[1] if query.ndim == 2 and len(query) else (train.shape[1] if train.ndim == 2 else 8)
    nq, nt = len(query), len(train)
    idx = np.zeros(nq, np.int32)
    dist = np.zeros(nq, np.uint32)
    dist2 = np.zeros(nq, np.uint32)
    ctx.check(ctx.lib.pislam_match_hamming(ctx.h, words, ptr(query) if nq else None, nq, ptr(train) if nt else None, nt,
                                           ptr(idx) if nq else None, ptr(dist) if nq else None,
                                           ptr(dist2) if nq else None), "pislam_match_hamming")
    return idx, dist, dist2


def matchHammingBatch(qdesc, qcounts, tdesc, tcounts, idx=None, dist=None, dist2=None, *, ctx: Context | None = None):
    """Batched matcher on device-resident front-end outputs (torch tensors [batch][max_kp][words] and
    [batch] counts, as OrbFrontend writes them): pair b matches qdesc[b] against tdesc[b].  Returns
    (idx int32, dist int32, dist2 int32) tensors [batch][max_kp]; asynchronous on the ctx stream."""
    import torch
    ctx = ctx or default_context()
    batch, qs, words = qdesc.shape
    ts = tdesc.shape[1]
    if idx is None:
        idx = torch.empty((batch, qs), dtype=torch.int32, device=qdesc.device)
    if dist is None:
        dist = torch.empty((batch, qs), dtype=torch.int32, device=qdesc.device)
    if dist2 is None:
        dist2 = torch.empty((batch, qs), dtype=torch.int32, device=qdesc.device)
    ctx.check(ctx.lib.pislam_match_hamming_batch(ctx.h, words, ptr(qdesc), ptr(qcounts), qs, ptr(tdesc), ptr(tcounts), ts,
                                                 batch, ptr(idx), ptr(dist), ptr(dist2)), "pislam_match_hamming_batch")
    return idx, dist, dist2


# ---- Gaussian.h:48, Bilinear.h:42, Bilinear.h:165 -------------------------------------
def gaussian5x5(width, height, img, out, *, ctx: Context | None = None):
    """pislam::gaussian5x5<vstep>(width, height, img, out); img may be out (in place)."""
    ctx = ctx or default_context()
    ctx.check(ctx.lib.pislam_gaussian5x5(ctx.h, _vstep(img), width, height, ptr(img), ptr(out)), "pislam_gaussian5x5")


def bilinear7_8(width, height, img, out, *, ctx: Context | None = None):
    """pislam::bilinear7_8<vstep>(width, height, img, out)."""
    ctx = ctx or default_context()
    ctx.check(ctx.lib.pislam_bilinear7_8(ctx.h, _vstep(img), width, height, ptr(img), ptr(out)), "pislam_bilinear7_8")


def bilinear13_16(width, height, img, out, *, ctx: Context | None = None):
    """pislam::bilinear13_16<vstep>(width, height, img, out)."""
    ctx = ctx or default_context()
    ctx.check(ctx.lib.pislam_bilinear13_16(ctx.h, _vstep(img), width, height, ptr(img), ptr(out)),
              "pislam_bilinear13_16")


# ---- on-GPU pyramid build (BASELINE config 5) ------------------------------------------
DEFAULT_CHAIN = (2, 1, 2, 2, 1, 2, 2)     # 13/16, 7/8, 13/16, ... : (13/16)^2 * 7/8 = 0.578 ~ 1.2^-3


class PyramidBuilder:
    """frames uint8 [batch][h][w] (device) -> stacked pyramids uint8 [batch][rows][vstep] (device):
    level 0 = gaussian5x5(frame), level k+1 = bilinear13_16 / bilinear7_8 of level k."""

    def __init__(self, width: int, height: int, steps=DEFAULT_CHAIN, *, blur=True, vstep_min=0,
                 ctx: Context | None = None):
        self.ctx = ctx or default_context()
        self.nlevels = len(steps) + 1
        self.steps = (ctypes.c_int32 * max(1, len(steps)))(*steps)
        self.levels_c = (Level * self.nlevels)()
        vs, rows = ctypes.c_int32(0), ctypes.c_int32(0)
        rc = self.ctx.lib.pislam_pyramid_layout(width, height, self.nlevels, self.steps, vstep_min, self.levels_c,
                                                ctypes.byref(vs), ctypes.byref(rows))
        if rc != 0:
            raise capi.PislamError(f"pislam_pyramid_layout failed ({rc})")
        self.vstep, self.rows, self.blur = vs.value, rows.value, bool(blur)
        self.levels = [(l.width, l.height, l.row0, l.col0) for l in self.levels_c]

    def __call__(self, frames, pyramids, margins_clean: bool = False):
        """margins_clean: the caller vouches that `pyramids` was last filled by this builder with this layout and
        not written to since (PISLAM_BUILD_MARGINS_CLEAN) — the zero margins are then not re-established."""
        c = self.ctx
        batch = int(frames.shape[0])
        flags = (1 if self.blur else 0) | (2 if margins_clean else 0)
        c.check(c.lib.pislam_pyramid_build_batch(c.h, self.nlevels, self.steps, self.levels_c, ptr(frames),
                                                 int(frames.shape[2]), int(frames.shape[1]) * int(frames.shape[2]),
                                                 batch, ptr(pyramids), self.vstep, self.rows, self.rows * self.vstep,
                                                 flags), "pislam_pyramid_build_batch")


# ---- the measured path ---------------------------------------------------------
class OrbFrontend:
    """Batch of device-resident stacked pyramids -> keypoints + descriptors + counts.

    Runs the call sequence of reference demo/demo.cpp:77-101 for every pyramid on the
    GPU (pislam_orb_frontend_batch).  All tensors are torch device tensors."""

    def __init__(self, levels, vstep: int, rows: int, *, border=16, fast_threshold=20,
                 harris_threshold=1 << 15, log_bucket_size=0, bucket_limit=5, words=8,
                 max_keypoints=4096, ctx: Context | None = None):
        self.ctx = ctx or default_context()
        lv = []
        for t in levels:
            w, h, r0 = t[0], t[1], t[2]
            c0 = t[3] if len(t) > 3 else 0
            lv.append(Level(w, h, r0, c0))
        self.levels = (Level * len(lv))(*lv)
        self.params = FrontendParams(vstep, rows, len(lv), border, fast_threshold, harris_threshold,
                                     log_bucket_size, bucket_limit, words, max_keypoints)

    def reserve(self, batch: int):
        c = self.ctx
        c.check(c.lib.pislam_frontend_reserve(c.h, ctypes.byref(self.params), self.levels, batch),
                "pislam_frontend_reserve")

    def alloc_outputs(self, batch: int, device):
        import torch
        p = self.params
        kp = torch.zeros((batch, p.max_keypoints), dtype=torch.int32, device=device)
        desc = torch.zeros((batch, p.max_keypoints, p.words), dtype=torch.int32, device=device)
        counts = torch.zeros((batch,), dtype=torch.int32, device=device)
        return kp, desc, counts

    def __call__(self, pyramids, kp, desc, counts):
        """pyramids: uint8 [batch][rows][vstep] device tensor; outputs int32 device tensors
        (bit patterns are the reference's uint32)."""
        c = self.ctx
        batch = int(pyramids.shape[0])
        stride = int(pyramids.stride(0)) if hasattr(pyramids, "stride") else self.params.rows * self.params.vstep
        c.check(c.lib.pislam_orb_frontend_batch(c.h, ctypes.byref(self.params), self.levels, ptr(pyramids),
                                                stride, batch, ptr(kp), ptr(desc), ptr(counts)),
                "pislam_orb_frontend_batch")

    def score_map(self, b: int) -> np.ndarray:
        c = self.ctx
        out = np.zeros((self.params.rows, self.params.vstep), np.uint8)
        c.check(c.lib.pislam_frontend_get_score_map(c.h, b, ptr(out)), "pislam_frontend_get_score_map")
        return out

    def last_timing(self):
        c = self.ctx
        tot = ctypes.c_float(0)
        st = (ctypes.c_float * 3)()
        c.check(c.lib.pislam_frontend_last_timing(c.h, ctypes.byref(tot), ctypes.byref(st)),
                "pislam_frontend_last_timing")
        return float(tot.value), [float(v) for v in st]

    PATH_STAGED, PATH_FUSED, PATH_ONE_LAUNCH, PATH_BUCKET_SELECT, PATH_BUCKETS_IN_STRIPS, PATH_GENERIC_ORB = 1, 2, 4, 8, 16, 32
    PATH_FRAME_TIMED_OUT = 64          # the one-launch path timed out earlier on this context: three launches from then on
    COUNT_INVALID = 0xFFFFFFFF         # PISLAM_COUNT_INVALID: counts[i] of a pyramid a timed-out one-launch call did not produce

    def last_path_of(self, ctx) -> int:
        """last_path() of another context (a pipeline lane) that ran this front-end's parameters."""
        return int(ctx.lib.pislam_frontend_last_path(ctx.h))

    def last_path(self) -> int:
        """Bit mask PATH_* of the path the last call on this context took — see pislam_frontend_last_path."""
        return int(self.ctx.lib.pislam_frontend_last_path(self.ctx.h))

    def last_stats(self):
        """(strips redone by the overflow pass, strips) of the last call — see pislam_frontend_last_stats."""
        c = self.ctx
        st = (ctypes.c_uint32 * 2)()
        c.check(c.lib.pislam_frontend_last_stats(c.h, ctypes.byref(st)), "pislam_frontend_last_stats")
        return int(st[0]), int(st[1])
