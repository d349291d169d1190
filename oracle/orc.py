"""oracle/orc.py — TEST INFRASTRUCTURE: ctypes binding of oracle/liborc.so.

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import
this module.  The product (pislam_amd) never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_u8p = ctypes.c_void_p
_i, _sz = ctypes.c_int, ctypes.c_size_t


def build(force=False):
    """Compile liborc.so (and oracle/_ref when /root/reference is present)."""
    so = os.path.join(_HERE, "liborc.so")
    src = os.path.join(_HERE, "pislam_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liborc.so"], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/include/Brief.h"):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.orc_fast_detect.argtypes = [_i, _i, _i, _i, c_u8p, c_u8p, _i]
        L.orc_fast_detect.restype = None
        L.orc_fast_score_harris.argtypes = [_i, _i, _i, _i, c_u8p, ctypes.c_int32, c_u8p]
        L.orc_fast_score_harris.restype = None
        L.orc_harris_score_sobel.argtypes = [_i, c_u8p, _i, _i, ctypes.c_int32]
        L.orc_harris_score_sobel.restype = ctypes.c_uint8
        L.orc_harris_eval.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int32, ctypes.c_int32]
        L.orc_harris_eval.restype = ctypes.c_uint8
        L.orc_fast_extract.argtypes = [_i, _i, _i, _i, _i, _i, c_u8p, ctypes.c_void_p, _sz]
        L.orc_fast_extract.restype = _sz
        L.orc_centroids_size.argtypes = [_sz]
        L.orc_centroids_size.restype = _sz
        L.orc_orb_centroids.argtypes = [_i, c_u8p, ctypes.c_void_p, _sz, ctypes.c_void_p]
        L.orc_orb_centroids.restype = None
        L.orc_atan2.argtypes = [ctypes.c_void_p, _sz, ctypes.c_void_p]
        L.orc_atan2.restype = None
        L.orc_angle_bin.argtypes = [ctypes.c_int32, ctypes.c_int32]
        L.orc_angle_bin.restype = ctypes.c_uint8
        L.orc_vrecpe.argtypes = [ctypes.c_float]
        L.orc_vrecpe.restype = ctypes.c_float
        L.orc_brief_table.argtypes = []
        L.orc_brief_table.restype = ctypes.POINTER(ctypes.c_int8)
        L.orc_brief_describe.argtypes = [_i, c_u8p, _i, _i, _i, _i, ctypes.c_void_p]
        L.orc_brief_describe.restype = None
        L.orc_orb_compute.argtypes = [_i, _i, c_u8p, ctypes.c_void_p, _sz, ctypes.c_void_p]
        L.orc_orb_compute.restype = None
        L.orc_pyramid.argtypes = [_i, _i, _i, ctypes.c_int32, _i, _i, _i, c_u8p, c_u8p,
                                  ctypes.c_void_p, _i, ctypes.c_void_p, ctypes.c_void_p, _sz,
                                  ctypes.c_void_p]
        L.orc_pyramid.restype = _sz
        for name in ("orc_gaussian5x5", "orc_bilinear7_8", "orc_bilinear13_16"):
            getattr(L, name).argtypes = [_i, _i, _i, c_u8p]
            getattr(L, name).restype = None
        L.orc_fill_spiral.argtypes = [_i, _i, _i, _i, _i, c_u8p]
        L.orc_fill_spiral.restype = None
        L.orc_fill_random.argtypes = [_i, _i, _i, c_u8p]
        L.orc_fill_random.restype = None
        L.orc_pyramid4.argtypes = L.orc_pyramid.argtypes
        L.orc_pyramid4.restype = _sz
        L.orc_pyramid_mt.argtypes = [_i, ctypes.c_double, _i, _i, _i, _i, ctypes.c_int32, _i, _i, _i, c_u8p, _sz, _i,
                                     ctypes.c_void_p, _i, _sz, ctypes.POINTER(ctypes.c_ulonglong),
                                     ctypes.POINTER(ctypes.c_double)]
        L.orc_pyramid_mt.restype = ctypes.c_ulonglong
        L.orc_match_hamming.argtypes = [_i, ctypes.c_void_p, _sz, ctypes.c_void_p, _sz, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_void_p]
        L.orc_match_hamming.restype = None
        _LIB = L
    return _LIB


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a


def fast_detect(img, out, width, height, threshold, border=16):
    """Fast.h:54 fastDetect<vstep,border>; img/out are 2-D uint8 [rows][vstep] level slices."""
    assert img.dtype == np.uint8 and out.dtype == np.uint8 and img.flags.c_contiguous and out.flags.c_contiguous
    lib().orc_fast_detect(img.shape[1], border, width, height, img.ctypes.data, out.ctypes.data, threshold)


def fast_score_harris(img, out, width, height, threshold=1 << 15, border=16):
    """Fast.h:166 fastScoreHarris<vstep,border>."""
    lib().orc_fast_score_harris(img.shape[1], border, width, height, img.ctypes.data, threshold, out.ctypes.data)


def fast_extract(out, width, height, border=16, log_bucket=0, bucket_limit=5):
    """Fast.h:196 fastExtract<vstep,border,logBucketSize,bucketLimit>; returns uint32 array."""
    cap = max(16, ((width + 1) // 2) * ((height + 1) // 2))
    dst = np.zeros(cap, np.uint32)
    n = lib().orc_fast_extract(out.shape[1], border, log_bucket, bucket_limit, width, height,
                               out.ctypes.data, dst.ctypes.data, cap)
    assert n <= cap
    return dst[:n].copy()


def orb_centroids(img, points):
    """Orb.h:80 orbCentroids<vstep>: int32, grouped [x0..x3,y0..y3]."""
    points = np.ascontiguousarray(points, np.uint32)
    n8 = lib().orc_centroids_size(len(points))
    cen = np.zeros(n8, np.int32)
    lib().orc_orb_centroids(img.shape[1], img.ctypes.data, points.ctypes.data, len(points), cen.ctypes.data)
    return cen


def atan2_bins(cen):
    """Orb.h:310 atan2: one uint8 per slot (padding included)."""
    cen = np.ascontiguousarray(cen, np.int32)
    ang = np.zeros(len(cen) // 2, np.uint8)
    lib().orc_atan2(cen.ctypes.data, len(cen), ang.ctypes.data)
    return ang


def vrecpe(x):
    return float(lib().orc_vrecpe(ctypes.c_float(x)))


def brief_table():
    p = lib().orc_brief_table()
    return np.ctypeslib.as_array(p, shape=(30, 256, 4)).copy()


def brief_describe(img, x, y, rot, words=8):
    out = np.zeros(words, np.uint32)
    lib().orc_brief_describe(img.shape[1], img.ctypes.data, x, y, rot, words, out.ctypes.data)
    return out


def orb_compute(img, points, words=8):
    """Orb.h:396 orbCompute<vstep,words>: uint32 [n][words]."""
    points = np.ascontiguousarray(points, np.uint32)
    desc = np.zeros((len(points), words), np.uint32)
    lib().orc_orb_compute(img.shape[1], words, img.ctypes.data, points.ctypes.data, len(points), desc.ctypes.data)
    return desc


def pyramid(img, levels, fast_threshold=20, harris_threshold=1 << 15, border=16,
            log_bucket=0, bucket_limit=5, words=8, cap=None, return_score=False):
    """demo.cpp:77-101 / README.md:67-82 call sequence over a stacked pyramid.

    levels: iterable of (width, height, row0).  Returns (kp uint32[n], desc uint32[n][words],
    level_counts uint32[nlevels]) and optionally the final score map."""
    assert img.dtype == np.uint8 and img.flags.c_contiguous
    lv = np.ascontiguousarray(np.array(levels, np.int32).reshape(-1, 3))
    if cap is None:
        cap = int(sum(((w + 1) // 2) * ((h + 1) // 2) for w, h, _ in lv.tolist()))
    score = np.zeros_like(img)
    kp = np.zeros(cap, np.uint32)
    desc = np.zeros((cap, words), np.uint32)
    lc = np.zeros(len(lv), np.uint32)
    n = lib().orc_pyramid(img.shape[1], border, fast_threshold, harris_threshold, log_bucket,
                          bucket_limit, words, img.ctypes.data, score.ctypes.data, lv.ctypes.data,
                          len(lv), kp.ctypes.data, desc.ctypes.data, cap, lc.ctypes.data)
    m = min(n, cap)
    res = (kp[:m].copy(), desc[:m].copy(), lc)
    return res + (score,) if return_score else res


def gaussian5x5(img, width, height):
    """test/GaussianTest.cpp:159-215 reference(): in place on a 2-D uint8 [rows][vstep] array."""
    assert img.dtype == np.uint8 and img.flags.c_contiguous
    lib().orc_gaussian5x5(img.shape[1], width, height, img.ctypes.data)


def bilinear7_8(img, width, height):
    """test/BilinearTest.cpp:171-196 reference7_8(): in place."""
    assert img.dtype == np.uint8 and img.flags.c_contiguous
    lib().orc_bilinear7_8(img.shape[1], width, height, img.ctypes.data)


def bilinear13_16(img, width, height):
    """test/BilinearTest.cpp:208-233 reference13_16(): in place."""
    assert img.dtype == np.uint8 and img.flags.c_contiguous
    lib().orc_bilinear13_16(img.shape[1], width, height, img.ctypes.data)


def fill_spiral(vstep, width, height, cx, cy, rows=None):
    """test/TestUtil.cpp:28-55 fill_spiral into a fresh [rows][vstep] buffer (rows >= height)."""
    buf = np.zeros((rows or height, vstep), np.uint8)
    lib().orc_fill_spiral(vstep, width, height, cx, cy, buf.ctypes.data)
    return buf


def fill_random(vstep, width, height, rows=None):
    """test/TestUtil.cpp:57-65 fill_random (default-seeded std::mt19937_64, one draw per pixel) into a fresh
    zeroed [rows][vstep] buffer."""
    buf = np.zeros((rows or height, vstep), np.uint8)
    lib().orc_fill_random(vstep, width, height, buf.ctypes.data)
    return buf


def levels4(levels):
    """(w, h, row0[, col0]) tuples -> contiguous int32 [n][4]."""
    return np.ascontiguousarray(np.array([(t[0], t[1], t[2], t[3] if len(t) > 3 else 0) for t in levels], np.int32))


def pyramid4(img, levels, fast_threshold=20, harris_threshold=1 << 15, border=16, log_bucket=0, bucket_limit=5,
             words=8, cap=None):
    """`pyramid` for levels placed anywhere in the buffer: (w, h, row0, col0) per level (packed layouts)."""
    assert img.dtype == np.uint8 and img.flags.c_contiguous
    lv = levels4(levels)
    if cap is None:
        cap = int(sum(((w + 1) // 2) * ((h + 1) // 2) for w, h, _, _ in lv.tolist()))
    score = np.zeros_like(img)
    kp = np.zeros(cap, np.uint32)
    desc = np.zeros((cap, words), np.uint32)
    lc = np.zeros(len(lv), np.uint32)
    n = lib().orc_pyramid4(img.shape[1], border, fast_threshold, harris_threshold, log_bucket, bucket_limit, words,
                           img.ctypes.data, score.ctypes.data, lv.ctypes.data, len(lv), kp.ctypes.data,
                           desc.ctypes.data, cap, lc.ctypes.data)
    m = min(n, cap)
    return kp[:m].copy(), desc[:m].copy(), lc


def pyramid_mt(imgs, levels, nthreads, min_seconds, fast_threshold=20, harris_threshold=1 << 15, border=16,
               log_bucket=0, bucket_limit=5, words=8, cap=16384):
    """The whole path on `nthreads` pthreads for >= min_seconds (timed in C): imgs uint8 [n][rows][vstep].
    Returns (keypoints, pyramids, seconds)."""
    assert imgs.dtype == np.uint8 and imgs.flags.c_contiguous and imgs.ndim == 3
    lv = levels4(levels)
    done, secs = ctypes.c_ulonglong(0), ctypes.c_double(0)
    kps = lib().orc_pyramid_mt(nthreads, float(min_seconds), imgs.shape[2], imgs.shape[1], border, fast_threshold,
                               harris_threshold, log_bucket, bucket_limit, words, imgs.ctypes.data,
                               imgs.shape[1] * imgs.shape[2], imgs.shape[0], lv.ctypes.data, len(lv), cap,
                               ctypes.byref(done), ctypes.byref(secs))
    return int(kps), int(done.value), float(secs.value)


def match_hamming(query, train):
    """Library-defined brute-force matcher (no reference counterpart): (idx int32, dist, dist2 uint32)."""
    query = np.ascontiguousarray(query, np.uint32)
    train = np.ascontiguousarray(train, np.uint32)
    words = query.shape[1] if query.ndim == 2 else train.shape[1]
    nq, nt = len(query), len(train)
    idx = np.zeros(nq, np.int32)
    dist = np.zeros(nq, np.uint32)
    dist2 = np.zeros(nq, np.uint32)
    lib().orc_match_hamming(words, query.ctypes.data, nq, train.ctypes.data, nt, idx.ctypes.data, dist.ctypes.data,
                            dist2.ctypes.data)
    return idx, dist, dist2
