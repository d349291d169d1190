"""Image preparation: gaussian5x5 / bilinear7_8 / bilinear13_16.

CPU part: the oracle's restatement of the reference tests' scalar references against independent
numpy formulations.  GPU part: the HIP kernels against the oracle over the reference tests' own
parameter ranges and fixtures (GaussianTest: 16..63 x 16..63 spiral, vstep 640; BilinearTest:
1..47 x 1..47 spiral + random, vstep 64)."""
import numpy as np
import pytest


def np_bilinear(a, w, h, N, M, f, skip):
    out = {}
    r = lambda v: (v + 128) >> 8
    smap = [i + sum(i > s for s in skip) for i in range(M)]
    for by in range(-(-h // N)):
        for bx in range(-(-w // N)):
            for y in range(M):
                for x in range(M):
                    p = a[by * N + smap[y]:by * N + smap[y] + 2, bx * N + smap[x]:bx * N + smap[x] + 2].astype(int)
                    h0 = r(p[0, 0] * f[x] + p[0, 1] * f[M - 1 - x])
                    h1 = r(p[1, 0] * f[x] + p[1, 1] * f[M - 1 - x])
                    out[(by * M + y, bx * M + x)] = r(h0 * f[y] + h1 * f[M - 1 - y])
    return out


def test_oracle_gaussian_equals_numpy_rhadd_tree(orc):
    from pislam_amd import synth
    rng = np.random.default_rng(0)
    for w, h in [(16, 16), (50, 37), (63, 17), (640, 480), (3, 3), (5, 4)]:
        a = rng.integers(0, 256, (h + 3, max(w + 5, 8)), dtype=np.uint8)
        b = a.copy()
        orc.gaussian5x5(b, w, h)
        assert (b[:h, :w] == synth.gaussian5x5(a[:h, :w])).all(), (w, h)
        assert (b[h:] == a[h:]).all() and (b[:, w:] == a[:, w:]).all()
    # a constant image stays constant, an impulse of 255 spreads to the binomial kernel (rounded)
    c = np.full((20, 24), 77, np.uint8); orc.gaussian5x5(c, 24, 20); assert (c == 77).all()
    imp = np.zeros((21, 24), np.uint8); imp[10, 12] = 255; orc.gaussian5x5(imp, 24, 21)
    assert imp[10, 12] == imp.max() and imp[10, 10] == imp[10, 14] and imp[8, 12] == imp[12, 12] and imp[7, 12] == 0


def test_oracle_bilinear_equals_pure_function_of_the_original(orc):
    """The reference's in-place loops (BilinearTest.cpp:171-233) equal an out-of-place evaluation on
    the original image; also the 13/16 filter bank keeps the reference's f[10] = 138."""
    rng = np.random.default_rng(1)
    f7 = [238, 201, 165, 128, 91, 55, 18]
    f13 = [226, 167, 108, 49, 246, 187, 128, 69, 10, 207, 138, 89, 30]
    for w, h in [(40, 33), (8, 8), (47, 1), (1, 47), (23, 31)]:
        a = rng.integers(0, 256, (64, 64), dtype=np.uint8)
        b = a.copy(); orc.bilinear7_8(b, w, h)
        for (oy, ox), v in np_bilinear(a, w, h, 8, 7, f7, []).items():
            assert b[oy, ox] == v
        b = a.copy(); orc.bilinear13_16(b, w, h)
        for (oy, ox), v in np_bilinear(a, w, h, 16, 13, f13, [3, 8]).items():
            assert b[oy, ox] == v
    flat = np.full((64, 64), 200, np.uint8); orc.bilinear7_8(flat, 48, 48)
    assert (flat[:42, :42] == 200).all()


@pytest.mark.gpu
def test_gaussian5x5_reference_test_matrix(gpu_ctx, orc):
    """GaussianTest.spiral/random over Range(16,64)^2 at vstep 640, in place, plus VGA / 720p."""
    from pislam_amd import frontend as fe
    rng = np.random.default_rng(2)
    sizes = [(w, h) for w in range(16, 64, 5) for h in range(16, 64, 7)] + [(63, 63), (16, 16), (640, 480), (1280, 720), (3, 3), (129, 17)]
    for w, h in sizes:
        vstep = 640 if w <= 640 else 1280
        for kind in ("spiral", "random", "rng"):
            if kind == "spiral":
                a = orc.fill_spiral(vstep, w, h, vstep // 3, vstep // 3, rows=h + 2)
            elif kind == "random":               # the reference's own fixture: test_util::fill_random (TestUtil.cpp:57)
                a = orc.fill_random(vstep, w, h, rows=h + 2)
            else:
                a = rng.integers(0, 256, (h + 2, vstep), dtype=np.uint8)
            exp = a.copy(); orc.gaussian5x5(exp, w, h)
            g = a.copy(); fe.gaussian5x5(w, h, g, g, ctx=gpu_ctx)              # in place, like the reference test
            assert (g == exp).all(), (w, h, kind, np.argwhere(g != exp)[:4])
            out = np.full_like(a, 9); fe.gaussian5x5(w, h, a, out, ctx=gpu_ctx)   # out of place: only w x h written
            assert (out[:h, :w] == exp[:h, :w]).all() and (out[h:] == 9).all() and (out[:, w:] == 9).all()


@pytest.mark.gpu
def test_bilinear_reference_test_matrix(gpu_ctx, orc):
    """BilinearTest.{spiral,random}{7_8,13_16} over Range(1,48)^2 at vstep 64, in place."""
    from pislam_amd import frontend as fe
    rng = np.random.default_rng(3)
    for w in list(range(1, 48, 3)) + [47, 16, 32]:
        for h in list(range(1, 48, 5)) + [47, 16, 32]:
            sp = orc.fill_spiral(64, w, h, 21, 21, rows=64)
            rd = orc.fill_random(64, w, h, rows=64)         # BilinearTest.cpp:95,149: test_util::fill_random
            rg = rng.integers(0, 256, (64, 64), dtype=np.uint8)
            for a in (sp, rd, rg):
                for name, N, M in (("bilinear7_8", 8, 7), ("bilinear13_16", 16, 13)):
                    exp = a.copy(); getattr(orc, name)(exp, w, h)
                    g = a.copy(); getattr(fe, name)(w, h, g, g, ctx=gpu_ctx)
                    assert (g == exp).all(), (name, w, h, np.argwhere(g != exp)[:4])
                    # the region the reference's tests assert on
                    oh, ow = h * M // N, w * M // N
                    out = np.zeros_like(a); getattr(fe, name)(w, h, a, out, ctx=gpu_ctx)
                    assert (out[:oh, :ow] == exp[:oh, :ow]).all()
    # VGA and a device-resident call
    import torch
    a = rng.integers(0, 256, (480, 640), dtype=np.uint8)
    for name in ("bilinear7_8", "bilinear13_16"):
        exp = a.copy(); getattr(orc, name)(exp, 640, 480)
        d_in = torch.from_numpy(a).cuda(); d_out = torch.zeros_like(d_in)
        getattr(fe, name)(640, 480, d_in, d_out, ctx=gpu_ctx)
        torch.cuda.synchronize()
        M, N = (7, 8) if name == "bilinear7_8" else (13, 16)
        assert (d_out.cpu().numpy()[:480 * M // N, :640 * M // N] == exp[:480 * M // N, :640 * M // N]).all()


def build_defined_mask(pb, steps):
    """Bytes a pyramid build defines: per level the rectangle it rewrites (level 0: the frame; level l > 0: the whole
    output blocks of the reduction into it) plus the zeroed margins its consumers read (32 columns to the right,
    16 rows below — include/pislam_hip.h, pislam_pyramid_build_batch).  Everything else is left untouched."""
    mask = np.zeros((pb.rows, pb.vstep), bool)
    for l, (w, h, r0, _) in enumerate(pb.levels):
        slot = (pb.levels[l + 1][2] if l + 1 < len(pb.levels) else pb.rows) - r0
        if l == 0:
            ww, wh = w, h
        else:
            N, M = (8, 7) if steps[l - 1] == 1 else (16, 13)
            pw, ph = pb.levels[l - 1][0], pb.levels[l - 1][1]
            ww, wh = -(-pw // N) * M, -(-ph // N) * M
        mask[r0:r0 + min(slot, wh + 16), :min(pb.vstep, ww + 32)] = True
    return mask


@pytest.mark.gpu
def test_pyramid_build_equals_reference_functions_in_sequence(gpu_ctx, orc):
    """config 5 builder: gaussian5x5 then the 13/16, 7/8 chain on a zeroed stacked buffer == the oracle
    running the reference tests' scalar references in the same order; then ORB on the built pyramid."""
    import torch
    from pislam_amd.frontend import OrbFrontend, PyramidBuilder
    from pislam_amd import synth
    rng = np.random.default_rng(9)
    for (w0, h0, steps) in [(1280, 720, (2, 1, 2, 2, 1, 2, 2)), (333, 251, (1, 2, 1)), (640, 480, (2, 2))]:
        pb = PyramidBuilder(w0, h0, steps, ctx=gpu_ctx)
        B = 3
        frames = np.stack([synth.make_level0(50 + i, w0, h0) if i else rng.integers(0, 256, (h0, w0), dtype=np.uint8)
                           for i in range(B)])
        d_fr = torch.from_numpy(frames).cuda()
        # a dirty buffer: the build must define everything its consumers read and leave the rest alone
        d_pyr = torch.full((B, pb.rows, pb.vstep), 0xAB, dtype=torch.uint8, device="cuda")
        pb(d_fr, d_pyr)
        torch.cuda.synchronize()
        got = d_pyr.cpu().numpy()
        mask = build_defined_mask(pb, steps)
        assert (got[:, ~mask] == 0xAB).all()
        # refill of the same buffer with the margins vouched for (PISLAM_BUILD_MARGINS_CLEAN): identical bytes
        snap = d_pyr.clone()
        d_pyr[:, :, :] = torch.where(torch.from_numpy(mask).cuda()[None], d_pyr, torch.zeros_like(d_pyr) + 0xCD)
        lvl_only = torch.zeros_like(d_pyr, dtype=torch.bool)
        pb(d_fr, d_pyr, margins_clean=True)
        torch.cuda.synchronize()
        assert torch.equal(torch.where(torch.from_numpy(mask).cuda()[None], d_pyr, snap), snap) and not lvl_only.any()
        got = np.where(mask[None], got, 0)
        for b in range(B):
            exp = np.zeros((pb.rows, pb.vstep), np.uint8)
            w, h, r0, _ = pb.levels[0]
            exp[r0:r0 + h, :w] = frames[b]
            lvl = exp[r0:]
            orc.gaussian5x5(lvl, w, h)
            for k, st in enumerate(steps):
                w, h, r0, _ = pb.levels[k]
                tmp = exp[r0:].copy()                       # out-of-place: level k -> level k+1 slot
                (orc.bilinear7_8 if st == 1 else orc.bilinear13_16)(tmp, w, h)
                w1, h1, r1, _ = pb.levels[k + 1]
                N, M = (8, 7) if st == 1 else (16, 13)
                oh, ow = -(-h // N) * M, -(-w // N) * M      # whole blocks, like the reference loops
                exp[r1:r1 + oh, :ow] = tmp[:oh, :ow]
                assert (w1, h1) == (w * M // N, h * M // N)
            assert (got[b] == exp).all(), (w0, h0, b, np.argwhere(got[b] != exp)[:4])
        # ORB front-end on the built pyramids == oracle on the same buffers
        fe = OrbFrontend(pb.levels, vstep=pb.vstep, rows=pb.rows, max_keypoints=8192, ctx=gpu_ctx)
        kp, desc, counts = fe.alloc_outputs(B, d_pyr.device)
        fe(d_pyr, kp, desc, counts)
        torch.cuda.synchronize()
        c = counts.cpu().numpy().view(np.uint32); k = kp.cpu().numpy().view(np.uint32); d = desc.cpu().numpy().view(np.uint32)
        for b in (1, 2):
            okp, odesc, _ = orc.pyramid(got[b], [l[:3] for l in pb.levels])
            assert c[b] == len(okp) and (k[b, :len(okp)] == okp).all() and (d[b, :len(okp)] == odesc).all()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1280, 720, (2, 1, 2, 2, 1, 2, 2), 64), (640, 480, (2, 1, 2, 2, 1, 2, 2), 5), (333, 251, (1, 2, 1), 3),
                                   (1920, 1080, (2, 2, 1, 2), 2)])
def test_one_launch_build_equals_one_launch_per_level(gpu_ctx, shape):
    """pp::k_bilinear_chain (every reduction of a build in ONE launch, row bands of level l handed to the workgroups of
    level l + 1 through counters) writes the bytes of the one-launch-per-level build (option "build_chain" 0) — at the bench's
    batch of 64 720p frames too — on frames that CHANGE between builds into the same buffer (a workgroup that started
    before its source rows were complete, or read them from a stale cache line, would show the previous frames' pixels), and a
    launch re-arms its own counters (five builds in a row, no host-side reset)."""
    import torch
    from pislam_amd.frontend import PyramidBuilder
    w0, h0, steps, B = shape
    rng = np.random.default_rng(w0 * 7 + B)
    pb = PyramidBuilder(w0, h0, steps, ctx=gpu_ctx)
    d_a = torch.zeros((B, pb.rows, pb.vstep), dtype=torch.uint8, device="cuda")
    d_b = torch.zeros_like(d_a)
    for rep in range(5):
        base = rng.integers(0, 256, (min(B, 4), h0, w0), dtype=np.uint8)
        frames = torch.from_numpy(base).cuda()[torch.arange(B, device="cuda") % base.shape[0]].contiguous()
        if B > 4:
            frames[B // 2:] = 255 - frames[B // 2:]          # (the second half of a large batch differs from the first)
        gpu_ctx.set_option("build_chain", 1)
        pb(frames, d_a, margins_clean=rep > 0)
        gpu_ctx.set_option("build_chain", 0)
        pb(frames, d_b, margins_clean=rep > 0)
        torch.cuda.synchronize()
        assert torch.equal(d_a, d_b), (shape, rep, torch.nonzero(d_a != d_b)[:4])


@pytest.mark.gpu
@pytest.mark.parametrize("hook", [1 | (12 << 8), 2], ids=["a-band-never-completes", "a-frame-meets-two-XCDs"])
def test_one_launch_build_faults_are_reported_and_the_context_recovers(hook):
    """The build chain's two safety nets (test hook "frame_test"): bit 0 — the first workgroup of level 1 never publishes its
    rows, the poll limit is 2^12: a bounded wait expires; bit 1 — one workgroup pretends to run on another XCD than the
    frame's first workgroup: the one-XCD-per-frame placement the hand-over through the L2 rests on is violated.  Either way
    the next call on the context fails once naming the fault, after that the context builds one launch per level and the
    result is the reference's again."""
    import torch
    from pislam_amd.capi import Context, PislamError
    from pislam_amd.frontend import PyramidBuilder
    ctx = Context(device=0)
    pb = PyramidBuilder(640, 480, (2, 1, 2, 2), ctx=ctx)
    rng = np.random.default_rng(3)
    frames = torch.from_numpy(rng.integers(0, 256, (4, 480, 640), dtype=np.uint8)).cuda()
    good = torch.zeros((4, pb.rows, pb.vstep), dtype=torch.uint8, device="cuda")
    pyr = torch.zeros_like(good)
    ctx.set_option("build_chain", 0)
    pb(frames, good)
    ctx.set_option("build_chain", 1)
    pb(frames, pyr)
    ctx.synchronize()
    assert torch.equal(pyr, good)
    ctx.set_option("frame_test", hook)
    pb(frames, pyr)
    ctx.set_option("frame_test", 0)
    with pytest.raises(PislamError, match="one-launch pyramid build"):
        ctx.synchronize()
    for rep in range(2):
        pyr.zero_()
        pb(frames, pyr)
        ctx.synchronize()
        assert torch.equal(pyr, good)
    ctx.set_option("frame_rearm", 1)                    # the one-launch build again (its counters were reset)
    for rep in range(2):
        pyr.zero_()
        pb(frames, pyr)
        ctx.synchronize()
        assert torch.equal(pyr, good)
    ctx.close()


@pytest.mark.gpu
def test_pyramid_build_flags_are_checked(gpu_ctx):
    """ABI 2: the last argument of pislam_pyramid_build_batch is a PISLAM_BUILD_* bitmask (ABI 1: `blur`, any non-zero
    value): unknown bits are refused instead of being read as flags, and PISLAM_BUILD_CHECK_MARGINS verifies a
    caller's PISLAM_BUILD_MARGINS_CLEAN promise (dirty margins -> PISLAM_ERR_INVALID, clean ones pass)."""
    import ctypes
    import torch
    from pislam_amd import capi
    from pislam_amd.frontend import PyramidBuilder
    pb = PyramidBuilder(320, 240, (2, 1), ctx=gpu_ctx)
    B = 2
    d_fr = torch.randint(0, 256, (B, 240, 320), dtype=torch.uint8, device="cuda")
    d_pyr = torch.full((B, pb.rows, pb.vstep), 0x5A, dtype=torch.uint8, device="cuda")

    def build(flags):
        return gpu_ctx.lib.pislam_pyramid_build_batch(gpu_ctx.h, pb.nlevels, pb.steps, pb.levels_c, capi.ptr(d_fr), 320, 240 * 320, B,
                                                      capi.ptr(d_pyr), pb.vstep, pb.rows, pb.rows * pb.vstep, flags)
    for bad in (8, -1, 1 | 16, 0x100):
        assert build(bad) == -1                                  # PISLAM_ERR_INVALID, nothing launched
    assert build(1 | 2 | 4) == -1                                # the buffer is dirty (0x5A everywhere): the promise is false
    assert b"margins" in gpu_ctx.lib.pislam_last_error(gpu_ctx.h)
    assert build(1) == 0                                         # a normal build establishes the margins ...
    torch.cuda.synchronize()
    assert build(1 | 2 | 4) == 0                                 # ... and now the promise holds
    torch.cuda.synchronize()
