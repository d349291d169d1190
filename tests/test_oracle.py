"""CPU suite: the oracle against the reference's pins (no GPU, no /root/reference needed)."""
import ctypes
import os

import numpy as np
import pytest

from conftest import GOLDEN, SURVEY_PINS, sha16

RING = [(-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3), (0, 3), (1, 3), (2, 2),
        (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2)]   # (dy, dx)
UMAX = [15, 15, 15, 15, 15, 15, 14, 14, 13, 13, 12, 11, 10, 9, 7, 5]


def test_golden_fixture_matches_survey_pins(demo):
    for k, v in SURVEY_PINS.items():
        assert sha16(demo[k]) == v, k


def test_oracle_reproduces_reference_outputs_on_demo_pyramid(demo, orc):
    """Every stage of the oracle on the reference's demo input == the hashes recorded from the
    reference's own headers: det map, score map, both keypoint lists incl. order,
    centroids, angle bins, descriptors."""
    img, levels = demo["img"], demo["levels"]
    det = np.zeros_like(img)
    for w, h, r0 in levels:
        orc.fast_detect(img[r0:r0 + h], det[r0:r0 + h], w, h, 20)
    assert sha16(det) == SURVEY_PINS["det"]
    score = det.copy()
    for w, h, r0 in levels:
        orc.fast_score_harris(img[r0:r0 + h], score[r0:r0 + h], w, h)
    assert sha16(score) == SURVEY_PINS["score"]
    kp = np.concatenate([orc.fast_extract(score[r0:r0 + h], w, h) + np.uint32(r0) for w, h, r0 in levels])
    assert len(kp) == 1754 and sha16(kp.astype(np.uint32)) == SURVEY_PINS["kp"]
    kpb = np.concatenate([orc.fast_extract(score[r0:r0 + h], w, h, log_bucket=4, bucket_limit=3) + np.uint32(r0)
                          for w, h, r0 in levels])
    assert len(kpb) == 1315 and sha16(kpb.astype(np.uint32)) == SURVEY_PINS["kp_bucket43"]
    cen = orc.orb_centroids(img, kp)
    assert len(cen) == 3512 and sha16(cen) == SURVEY_PINS["centroids"]
    ang = orc.atan2_bins(cen)
    assert len(ang) == 1756 and sha16(ang) == SURVEY_PINS["angles"]
    desc = orc.orb_compute(img, kp)
    assert desc.size == 14032 and sha16(desc) == SURVEY_PINS["desc"]
    assert int(kp[0]) == 0x891ED010 and int(desc[0, 0]) == 0xB3BFD04F
    # whole-pyramid driver == stage-by-stage
    k2, d2, lc = orc.pyramid(img, levels)
    assert (k2 == kp).all() and (d2 == desc).all()
    assert lc.tolist() == [271, 296, 275, 255, 229, 184, 148, 96]


def test_vrecpe_known_answers(orc):
    """ARM ARM FPRecipEstimate known answers."""
    assert orc.vrecpe(1.0) == 0.998046875
    assert orc.vrecpe(2.0) == 0.4990234375
    assert orc.vrecpe(3.0) == 0.3330078125
    assert orc.vrecpe(0.0) == float("inf")
    assert orc.vrecpe(float("inf")) == 0.0
    assert orc.vrecpe(-4.0) == -orc.vrecpe(4.0)
    # integer form == the pseudocode's floating form for every 8-bit mantissa bucket
    for q in range(256, 512):
        r = 1.0 / ((q + 0.5) / 512.0)
        s = int(256.0 * r + 0.5)
        assert ((1 << 19) + (2 * q + 1)) // (2 * (2 * q + 1)) == s


def test_angle_bin_quadrants_and_padding(orc):
    L = orc.lib()
    assert L.orc_angle_bin(0, 0) == 7            # the (0,0) padding slots
    assert L.orc_angle_bin(1000, 0) == 0
    assert L.orc_angle_bin(0, 1000) == 7
    assert L.orc_angle_bin(-1000, 0) == 15
    assert L.orc_angle_bin(0, -1000) == 22
    # bins follow exact atan2 except near bin edges: compare away from edges
    rng = np.random.default_rng(1)
    bad = 0
    for _ in range(3000):
        x, y = (int(v) for v in rng.integers(-100000, 100000, 2))
        a = np.degrees(np.arctan2(y, x)) % 360.0
        if abs((a / 12.0) - round(a / 12.0)) < 0.05:
            continue
        bad += int(L.orc_angle_bin(x, y) != int(a // 12) % 30)
    assert bad == 0


def test_fast9_equals_textbook_definition(orc):
    """oracle's NEON-style clz/shift test == '9 contiguous ring pixels all brighter than c+t or
    all darker than c-t' on adversarial random images."""
    rng = np.random.default_rng(2)
    for w, h, t in [(48, 40, 20), (61, 37, 5), (64, 33, 60), (50, 50, 0), (52, 41, 255), (40, 40, 276)]:
        img = rng.integers(0, 256, (h, 96), dtype=np.uint8)
        img[:, : w // 2] = (img[:, : w // 2] // 64) * 64          # blocky half -> many corners
        out = np.zeros_like(img)
        B = 3
        orc.fast_detect(img, out, w, h, t, border=B)
        tt = t & 0xFF
        xend = B + 16 * -(-(w - 2 * B) // 16)
        for y in range(B, h - B):
            for x in range(B, xend):
                c = int(img[y, x])
                ring = [int(img[y + dy, x + dx]) for dy, dx in RING]
                br = [p > c + tt for p in ring]
                dk = [p < c - tt for p in ring]
                def arc(f):
                    f2 = f + f
                    return any(all(f2[s:s + 9]) for s in range(16))
                exp = 0xFF if (arc(br) or arc(dk)) else 0
                if w % 16 and x in (w, w + 1):
                    exp = 0
                assert out[y, x] == exp, (w, h, t, x, y)
        assert not out[:B].any() and not out[h - B:].any() and not out[:, :B].any()
        assert not out[:, max(xend, w + 2 if w % 16 else 0):].any()


def test_centroid_equals_bruteforce_circle(orc, demo):
    img = demo["img"]
    kp = demo["kp"][::97]
    cen = orc.orb_centroids(img, kp)
    for i, p in enumerate(kp):
        x, y = (int(p) >> 12) & 0xFFF, int(p) & 0xFFF
        m10 = m01 = 0
        for dy in range(-15, 16):
            u = UMAX[abs(dy)]
            row = img[y + dy, x - u:x + u + 1].astype(np.int64)
            m10 += int((row * np.arange(-u, u + 1)).sum())
            m01 += dy * int(row.sum())
        g = (i // 4) * 8 + (i % 4)
        assert cen[g] == m10 and cen[g + 4] == m01


def test_harris_eval_wraparound_and_encoding(orc):
    L = orc.lib()
    assert L.orc_harris_eval(0, 0, 0, 1 << 15) == 0
    # score = Ixx*Iyy - Ixy^2 - (Ixx+Iyy)^2/16
    for ixx, iyy, ixy in [(5000, 4000, 100), (30000, 30000, -20000), (900, 800, 0), (65535, 65535, 0)]:
        det = (ixx * iyy - ixy * ixy) & 0xFFFFFFFF
        tr = (((ixx + iyy) ** 2) & 0xFFFFFFFF) >> 4
        s = (det - tr) & 0xFFFFFFFF
        s = s - (1 << 32) if s >= (1 << 31) else s
        exp = 0
        if s > (1 << 15):
            exp = (np.float32(s).view(np.uint32) >> 20) & 0xFF
        assert L.orc_harris_eval(ixx, iyy, ixy, 1 << 15) == exp


def test_brief_table_equals_compiled_reference_probe(orc):
    """oracle's float32 rotation formula == table probed from the compiled reference Brief.h."""
    ref = np.load(os.path.join(GOLDEN, "brief_table_ref.npy"))
    tab = orc.brief_table()
    assert tab.shape == (30, 256, 4)
    assert (tab == ref).all()
    import hashlib
    assert hashlib.sha256(ref.tobytes()).hexdigest().startswith("15c108c2d4c62b94")
    assert tuple(ref[1, 0]) == (8, -1, 8, 7) and tuple(ref[7, 0]) == (4, 8, -4, 9)
    assert int((np.abs(ref) == 15).sum()) > 0


def test_brief_descriptor_equals_real_reference_when_built(orc, demo):
    """Direct execution of the real Brief.h (oracle/_ref/libbrief_ref.so, built in the dev container)."""
    so = os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "libbrief_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref not built here")
    ref = ctypes.CDLL(so)
    ref.ref_brief_describe_640.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    img = demo["img"]
    rng = np.random.default_rng(3)
    for i, p in enumerate(demo["kp"][::41]):
        x, y = (int(p) >> 12) & 0xFFF, int(p) & 0xFFF
        rot = int(rng.integers(0, 30))
        exp = np.zeros(8, np.uint32)
        ref.ref_brief_describe_640(img.ctypes.data, x, y, rot, exp.ctypes.data)
        assert (orc.brief_describe(img, x, y, rot) == exp).all()


def test_keypoint_codec_equals_real_reference_when_built():
    """Direct execution of the reference's include/Util.h (oracle/_ref/libutil_ref.so): the Python mirror of
    the codec and the decode helpers agree with it on edge and random values."""
    so = os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "libutil_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref not built here")
    from pislam_amd import frontend as fr
    ref = ctypes.CDLL(so)
    for f in ("ref_encodeFast", "ref_rencodeFastScore", "ref_decodeFastX", "ref_decodeFastY", "ref_decodeFastScore"):
        getattr(ref, f).restype = ctypes.c_uint32
        getattr(ref, f).argtypes = [ctypes.c_uint32] * (3 if f == "ref_encodeFast" else 2 if f == "ref_rencodeFastScore" else 1)
    rng = np.random.default_rng(11)
    vals = [(0, 0, 0), (255, 4095, 4095), (137, 493, 16), (1, 16, 2209)] + \
           [tuple(int(v) for v in rng.integers(0, [256, 4096, 4096])) for _ in range(200)]
    for sc, x, y in vals:
        e = ref.ref_encodeFast(sc, x, y)
        assert fr.encodeFast(sc, x, y) == e
        assert (fr.decodeFastX(e), fr.decodeFastY(e), fr.decodeFastScore(e)) == \
               (ref.ref_decodeFastX(e), ref.ref_decodeFastY(e), ref.ref_decodeFastScore(e)) == (x, y, sc)
        assert fr.rencodeFastScore((sc * 7) & 255, e) == ref.ref_rencodeFastScore((sc * 7) & 255, e)
    assert ref.ref_encodeFast(137, 493, 16) == 0x891ED010          # first keypoint of the demo image


def test_fill_spiral_equals_real_reference_when_built(orc):
    """Direct execution of the reference's own test fixture generator test/TestUtil.cpp:27
    (oracle/_ref/libtestutil_ref.so, built in the dev container) over the sizes GaussianTest / BilinearTest use."""
    so = os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "libtestutil_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref not built here")
    ref = ctypes.CDLL(so)
    ref.ref_fill_spiral.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p]
    cases = [(640, w, h, 640 // 3, 640 // 3) for w in (16, 31, 63) for h in (16, 40, 63)]      # GaussianTest.cpp
    cases += [(64, w, h, 21, 21) for w in (1, 7, 16, 33, 47) for h in (1, 8, 30, 47)]          # BilinearTest.cpp
    cases += [(640, 640, 480, 320, 240), (1280, 1280, 720, 600, 300)]
    for vstep, w, h, cx, cy in cases:
        # the reference marks every spiral point with row < vstep (its tests use vstep x vstep buffers);
        # the oracle only keeps the rows < height, which is all the parity tests read
        exp = np.full((max(vstep, h), vstep), 0xAA, np.uint8)
        ref.ref_fill_spiral(vstep, w, h, cx, cy, exp.ctypes.data)
        got = orc.fill_spiral(vstep, w, h, cx, cy, rows=h)
        assert (got == exp[:h]).all(), (vstep, w, h, cx, cy)


def test_fill_random_equals_real_reference_when_built(orc):
    """test_util::fill_random (TestUtil.cpp:57-65, the fixture of BilinearTest.random* / GaussianTest.random)
    executed directly from oracle/_ref against the oracle's mt19937_64 restatement."""
    so = os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "libtestutil_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref not built here")
    ref = ctypes.CDLL(so)
    if not hasattr(ref, "ref_fill_random"):
        pytest.skip("oracle/_ref predates ref_fill_random")
    ref.ref_fill_random.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p]
    for vstep, w, h in [(64, 1, 1), (64, 47, 47), (64, 13, 40), (640, 63, 63), (640, 16, 16), (640, 640, 480), (1280, 1280, 720)]:
        exp = np.zeros((h, vstep), np.uint8)
        ref.ref_fill_random(vstep, w, h, exp.ctypes.data)
        assert (orc.fill_random(vstep, w, h) == exp).all(), (vstep, w, h)
    # C++11 [rand.predef]: the 10000th draw of a default mt19937_64 is 9981545732273789042
    assert int(orc.fill_random(10000, 10000, 1)[0, 9999]) == 9981545732273789042 & 0xff


def test_pyramid4_packed_levels_equal_per_level_calls(orc):
    """orc_pyramid4 (levels anywhere in the buffer, BASELINE config 4's packed layout) against the per-level
    call sequence of README.md:67-82 with the level origin added (README.md:78)."""
    from pislam_amd import synth
    levels = synth.packed_level_table(320, 240)
    rows = synth.pyramid_rows(levels)
    img = synth.make_batch(5, 1, w0=320, h0=240, vstep=320, levels=levels, nshapes=40)[0]
    kp, desc, lc = orc.pyramid4(img, levels)
    assert any(t[3] != 0 for t in levels)
    exp = []
    for (w, h, r0, c0) in levels:
        sub = np.ascontiguousarray(img.reshape(-1)[r0 * 320 + c0:])
        pad = (-len(sub)) % 320
        sub = np.concatenate([sub, np.zeros(pad, np.uint8)]).reshape(-1, 320)
        out = np.zeros_like(sub)
        orc.fast_detect(sub, out, w, h, 20)
        orc.fast_score_harris(sub, out, w, h)
        exp.append(orc.fast_extract(out, w, h) + np.uint32((c0 << 12) | r0))
    assert lc.tolist() == [len(e) for e in exp]
    exp = np.concatenate(exp)
    assert (kp == exp).all() and len(kp) > 50
    want = np.zeros((len(exp), 8), np.uint32)
    orc.lib().orc_orb_compute(320, 8, img.ctypes.data, exp.ctypes.data, len(exp), want.ctypes.data)
    assert (desc == want).all()
    # the threaded driver of bench.py's cpu_baseline produces the same keypoint totals
    n, done, secs = orc.pyramid_mt(img[None], levels, 2, 0.05)
    assert done >= 2 and n == done * len(kp) and secs > 0 and rows == img.shape[0]


def test_extract_edge_cases(orc):
    rng = np.random.default_rng(4)
    # empty map, tiny level, ties, capacity clipping
    s = np.zeros((64, 64), np.uint8)
    assert len(orc.fast_extract(s, 40, 40)) == 0
    assert len(orc.fast_extract(s, 32, 32)) == 0          # width == 2*border: loops do not run
    s[20, 20] = s[20, 21] = s[21, 20] = s[21, 21] = 9      # 4-way tie inside one block -> v3 wins
    kp = orc.fast_extract(s, 50, 50)
    assert kp.tolist() == [(9 << 24) | (21 << 12) | 21]
    s[:] = 0
    s[18, 19] = 7
    s[18, 20] = 7                                          # tie across blocks: '>=' vs '>' rules
    kp = orc.fast_extract(s, 50, 50)
    assert kp.tolist() == [(7 << 24) | (20 << 12) | 18]   # left pixel loses: v1 needs ">" to its right, v0 only ">=" to its left
    dense = rng.integers(0, 256, (64, 64), dtype=np.uint8)
    full = orc.fast_extract(dense, 60, 60)
    dst = np.zeros(5, np.uint32)
    n = orc.lib().orc_fast_extract(64, 16, 0, 5, 60, 60, dense.ctypes.data, dst.ctypes.data, 5)
    assert n == len(full) and (dst == full[:5]).all()
    # buckets: each 16x16 cell keeps its `limit` largest packed words, ascending
    b = orc.fast_extract(dense, 60, 60, log_bucket=4, bucket_limit=2)
    cells = {}
    for v in full:
        x, y = (int(v) >> 12) & 0xFFF, int(v) & 0xFFF
        bx, by = ((x - 16) // 2 * 2) // 16, ((y - 16) // 2 * 2) // 16
        cells.setdefault((by, bx), []).append(int(v))
    exp = []
    for key in sorted(cells):
        exp += sorted(cells[key])[-2:]
    assert b.tolist() == exp


def test_synth_small_fixture_regression(synth_small, orc):
    from pislam_amd import synth
    lv = synth_small["levels"]
    p = synth.make_pyramid(7, w0=160, h0=120, nlevels=3, levels=lv, nshapes=12)
    assert (p == synth_small["img"]).all(), "synthetic generator changed"
    kp, desc, _ = orc.pyramid(p, lv)
    assert (kp == synth_small["kp"]).all() and (desc == synth_small["desc"]).all()


def _neon_harris_numpy(P, threshold):
    """harrisScoreSobel (reference Harris.h:80-248) restated at REGISTER level in numpy for N patches at once:
    every NEON intrinsic the reference issues is spelled out on arrays of 8 byte lanes (vhsub_u8 / vhadd_s8 with
    their widened floor-halving, the u64 lane shifts, vmull / vmlal wrapping in int16, the u16 / s16 pairwise
    widening adds, the 'rubbish' high word left out by Harris.h:226-239, harrisEval's mod-2^32 arithmetic and the
    float-bits byte).  Independent of oracle/pislam_oracle.c, which restates the same path in per-pixel scalar
    arithmetic.  P: uint8 [N][8][8] = rows y-3..y+4, columns x-3..x+4."""
    P = P.astype(np.int64)
    N = len(P)

    def s8(v):                     # reinterpret the low 8 bits as int8
        v = v & 0xFF
        return np.where(v >= 128, v - 256, v)

    def vhsub_u8(a, b):            # (a - b) >> 1 on the widened difference, low 8 bits kept
        return ((a - b) >> 1) & 0xFF

    def vhadd_s8(a, b):            # (a + b) >> 1 on the widened sum of signed lanes
        return s8((a + b) >> 1)

    def shr_u64(v, lanes):         # vshr_n_u64 by 8 * lanes bits: lane j <- lane j + lanes, zeros shifted in
        out = np.zeros_like(v)
        out[:, :8 - lanes] = v[:, lanes:]
        return out

    row = [P[:, r, :] for r in range(8)]
    dy = []
    for n0 in range(6):            # PISLAM_HARRIS_DY_SOBEL(n0, n0+1, n0+2)
        t1 = vhsub_u8(row[n0 + 2], row[n0])                 # bytes, then read as s8
        t2 = s8(shr_u64(t1, 2))
        d = s8(shr_u64(t1, 1))
        t1 = vhadd_s8(s8(t1), t2)
        dy.append(vhadd_s8(d, t1))
    dx = [s8(vhsub_u8(shr_u64(row[n], 2), row[n])) for n in range(8)]     # PISLAM_HARRIS_DX_SOBEL_1
    for n0 in range(6):            # PISLAM_HARRIS_DX_SOBEL_2(n0, n0+1, n0+2): overwrites dx[n0] only
        dx[n0] = vhadd_s8(dx[n0], dx[n0 + 2])
        dx[n0] = vhadd_s8(dx[n0], dx[n0 + 1])

    def s16(v):
        v = v & 0xFFFF
        return np.where(v >= 32768, v - 65536, v)

    xx32 = np.zeros((N, 4), np.int64)
    yy32 = np.zeros((N, 4), np.int64)
    xy32 = np.zeros((N, 4), np.int64)
    for n in (0, 2, 4):
        xx = s16(dx[n] * dx[n] + dx[n + 1] * dx[n + 1])      # vmull_s8 + vmlal_s8: int16 lanes wrap
        yy = s16(dy[n] * dy[n] + dy[n + 1] * dy[n + 1])
        xy = s16(dx[n] * dy[n] + dx[n + 1] * dy[n + 1])
        ux, uy = xx & 0xFFFF, yy & 0xFFFF                    # vreinterpretq_u16_s16
        xx32 = (xx32 + ux[:, 0::2] + ux[:, 1::2]) & 0xFFFFFFFF      # vpaddlq_u16 / vpadalq_u16
        yy32 = (yy32 + uy[:, 0::2] + uy[:, 1::2]) & 0xFFFFFFFF
        xy32 = xy32 + xy[:, 0::2] + xy[:, 1::2]                      # vpaddlq_s16 / vpadalq_s16 (no wrap possible: |.| < 2^18)
    # vpaddl of the LOW register's two words, lane 0 of (that + the high register): words 0 + 1 + 2; word 3 holds
    # the two rubbish columns and is never added
    Ixx = ((xx32[:, 0] + xx32[:, 1] + xx32[:, 2]) & 0xFFFFFFFF) >> 4
    Iyy = ((yy32[:, 0] + yy32[:, 1] + yy32[:, 2]) & 0xFFFFFFFF) >> 4
    Ixy = (xy32[:, 0] + xy32[:, 1] + xy32[:, 2]) >> 4                # vshr_n_s32: arithmetic
    M = 0xFFFFFFFF
    tr = (Ixx + Iyy) & M
    tr = ((tr * tr) & M) >> 4
    det = ((Ixx * Iyy) & M)
    det = (det - ((Ixy * Ixy) & M)) & M                      # vmls_s32 on the reinterpreted bits
    score = (det - tr) & M
    score = np.where(score >= (1 << 31), score - (1 << 32), score)   # vsub_s32
    bits = score.astype(np.float32).view(np.uint32).astype(np.int64)  # vcvt_f32_s32 (round to nearest even)
    return np.where(score > threshold, (bits >> 20) & 0xFF, 0).astype(np.uint8)


def test_harris_against_a_register_level_numpy_restatement_of_the_neon_code(orc):
    """>= 100 000 patches: uniform random, low-contrast, binary 0 / 255 (the saturating edge cases of the 8-bit
    halving arithmetic and the 16-bit product lanes: Harris.h:188-203), checkerboards, steps — several thresholds."""
    L = orc.lib()
    rng = np.random.default_rng(2024)
    parts = [rng.integers(0, 256, (60000, 8, 8), dtype=np.uint8),
             rng.choice(np.array([0, 255], np.uint8), (30000, 8, 8)),
             (128 + rng.integers(-12, 13, (10000, 8, 8))).astype(np.uint8),
             rng.choice(np.array([0, 1, 254, 255], np.uint8), (10000, 8, 8))]
    yy, xx = np.mgrid[0:8, 0:8]
    special = [((yy + xx) % 2 * 255).astype(np.uint8), ((yy + xx + 1) % 2 * 255).astype(np.uint8),
               ((xx >= 4) * 255).astype(np.uint8), ((yy >= 4) * 255).astype(np.uint8),
               ((xx % 2) * 255).astype(np.uint8), ((yy % 2) * 255).astype(np.uint8),
               np.zeros((8, 8), np.uint8), np.full((8, 8), 255, np.uint8),
               (((yy >= 3) & (xx >= 3)) * 255).astype(np.uint8), (((yy // 2 + xx // 2) % 2) * 255).astype(np.uint8)]
    P = np.concatenate(parts + [np.stack(special)])
    assert len(P) >= 100000
    # the oracle reads patches out of an image: lay the patches side by side, 16 rows x (8 N) columns of context
    vstep = 8 * 512
    for thr in (1 << 15, 0, -(1 << 31), 1 << 22):
        want = _neon_harris_numpy(P, thr)
        got = np.zeros(len(P), np.uint8)
        for c0 in range(0, len(P), 512):
            blk = P[c0:c0 + 512]
            img = np.zeros((8, vstep), np.uint8)
            img[:, :8 * len(blk)] = blk.transpose(1, 0, 2).reshape(8, -1)
            for i in range(len(blk)):
                got[c0 + i] = L.orc_harris_score_sobel(vstep, img.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), 8 * i + 3, 3, thr)
        bad = np.flatnonzero(got != want)
        assert len(bad) == 0, (thr, bad[:5], got[bad[:5]], want[bad[:5]])
        if thr == 1 << 15:
            assert (want != 0).sum() > 10000            # the comparison is not vacuous


def test_parallel_synthetic_generator_equals_the_serial_one():
    """bench.py generates its input (one batch per pipeline lane) with worker processes (pislam_amd.synth.make_many):
    the bytes must be those of make_pyramid / make_level0 for every index, in the order asked for."""
    from pislam_amd import synth
    idx = [7, 3, 11, 5, 300, 301, 2, 9, 1000]
    a = synth.make_many(idx, workers=3, w0=160, h0=120, nlevels=3)
    b = np.stack([synth.make_pyramid(i, w0=160, h0=120, nlevels=3) for i in idx])
    assert a.shape == b.shape and (a == b).all()
    f = synth.make_many(idx, workers=3, kind="level0", w0=96, h0=64)
    assert (f == np.stack([synth.make_level0(i, 96, 64) for i in idx])).all()
    lv = synth.packed_level_table(320, 240)
    p = synth.make_many(idx[:6], workers=2, w0=320, h0=240, vstep=320, levels=lv)
    assert p.shape == (6, synth.pyramid_rows(lv), 320)
    assert (p[0] == synth.make_pyramid(idx[0], w0=320, h0=240, vstep=320, levels=lv)).all()
