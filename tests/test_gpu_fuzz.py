"""Seeded configuration fuzzing of the batch path against the oracle: random level tables, borders,
thresholds (incl. extreme ones), bucket modes, descriptor widths, capacities, input statistics (noise,
blocky, blocky + noise) and every tuning option (pipeline, layout, run length, strip height, x-tiles).
The same generator, run over 4000 configurations during development, found the two scan-fallback
buffer-size bugs that only show with narrow x-tiles."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_case(seed):
    rng = np.random.default_rng(seed)
    nl = int(rng.integers(1, 6))
    vstep = int(rng.choice([192, 208, 256, 320]))
    levels, row = [], 0
    for _ in range(nl):
        w, h = int(rng.integers(40, vstep + 1)), int(rng.integers(40, 160))
        levels.append((w, h, row))
        row += h
    rows = row + int(rng.integers(0, 3))
    batch = int(rng.integers(1, 6))
    kind = int(rng.integers(0, 3))
    if kind == 0:
        pyr = rng.integers(0, 256, (batch, rows, vstep), dtype=np.uint8)
    else:
        base = rng.integers(0, 256, (batch, rows // 4 + 2, vstep // 4 + 2), dtype=np.uint8)
        pyr = np.kron(base, np.ones((4, 4), np.uint8))[:, :rows, :vstep].copy()
        if kind == 2:
            pyr = (pyr.astype(np.int32) + rng.integers(-6, 7, pyr.shape)).clip(0, 255).astype(np.uint8)
    par = dict(border=int(rng.integers(16, 23)), fast_threshold=int(rng.choice([0, 5, 20, 40, 120, 255])),
               harris_threshold=int(rng.choice([-(1 << 31), -1000, 0, 1 << 10, 1 << 15, 1 << 22, (1 << 31) - 1])),
               log_bucket_size=int(rng.choice([0, 0, 2, 3, 4, 5])), bucket_limit=int(rng.integers(1, 7)),
               words=int(rng.choice([1, 2, 4, 8])), max_keypoints=int(rng.choice([16, 300, 4096])))
    opts = dict(pipeline=int(rng.choice([0, 1, 2, 2])), alias=int(rng.integers(0, 2)), run_len=int(rng.choice([0, 1, 3, 9])),
                strip_rows=int(rng.choice([0, 0, 10, 16, 22, 32])), sub_batches=1 + int(rng.choice([0, 0, 0, 64, 96])) // 48,
                orb_in_strip=int(rng.integers(0, 2)), tile_cols=int(rng.choice([0, -1, 64, 96, 160])),
                strip_rows_max=int(rng.choice([0, 0, 36, 56, 64])), bucket_select=int(rng.choice([1, 1, 0])))
    opts["frame"] = int(rng.choice([8, 1, 0]))      # (drawn last: the configurations of earlier campaigns keep their seeds)
    return levels, vstep, rows, pyr, par, opts


@pytest.mark.parametrize("chunk", range(6))
def test_random_configurations_match_the_oracle(gpu_ctx, orc, chunk):
    import torch
    from pislam_amd.frontend import OrbFrontend
    dev = torch.device("cuda:0")
    try:
        for seed in range(20000 + 25 * chunk, 20000 + 25 * (chunk + 1)):
            levels, vstep, rows, pyr, par, opts = make_case(seed)
            for k, v in opts.items():
                gpu_ctx.set_option(k, v)
            fe = OrbFrontend(levels, vstep=vstep, rows=rows, ctx=gpu_ctx, **par)
            kp, desc, counts = fe.alloc_outputs(len(pyr), dev)
            fe(torch.from_numpy(pyr).to(dev), kp, desc, counts)
            torch.cuda.synchronize()
            c = counts.cpu().numpy().view(np.uint32)
            k = kp.cpu().numpy().view(np.uint32)
            d = desc.cpu().numpy().view(np.uint32)
            for b in range(len(pyr)):
                okp, odesc, _ = orc.pyramid(pyr[b], levels, fast_threshold=par["fast_threshold"],
                                            harris_threshold=par["harris_threshold"], border=par["border"],
                                            log_bucket=par["log_bucket_size"], bucket_limit=par["bucket_limit"],
                                            words=par["words"])
                m = min(len(okp), par["max_keypoints"])
                assert c[b] == len(okp), (seed, b, par, opts, levels)
                assert (k[b, :m] == okp[:m]).all(), (seed, b, par, opts, levels)
                assert (d[b, :m].reshape(m, par["words"]) == odesc[:m]).all(), (seed, b, par, opts, levels)
    finally:
        for k, v in dict(pipeline=0, alias=1, run_len=0, strip_rows=0, sub_batches=1, orb_in_strip=0, tile_cols=0,
                         strip_rows_max=0, bucket_select=1, frame=1).items():
            gpu_ctx.set_option(k, v)


def test_random_single_level_calls_match_the_oracle(gpu_ctx, orc):
    """The four reference entry points on host arrays (any border >= 3 / 4 / 16, widths up to vstep —
    including the flat-addressing wrap of the over-classified columns —, stale `out` contents, bucket sizes
    1..64), the matcher and gaussian5x5 on random sizes."""
    from pislam_amd import frontend as pl
    for t in range(160):
        rng = np.random.default_rng(31000 + t)
        vstep = int(rng.choice([64, 96, 160, 208, 320, 640]))
        w, h = int(rng.integers(8, vstep + 1)), int(rng.integers(8, 200))
        rows = h + int(rng.integers(1, 4))
        kind = int(rng.integers(0, 3))
        if kind == 0:
            img = rng.integers(0, 256, (rows, vstep), dtype=np.uint8)
        else:
            base = rng.integers(0, 256, (rows // 4 + 2, vstep // 4 + 2), dtype=np.uint8)
            img = np.kron(base, np.ones((4, 4), np.uint8))[:rows, :vstep].copy()
            if kind == 2:
                img = (img.astype(np.int32) + rng.integers(-6, 7, img.shape)).clip(0, 255).astype(np.uint8)
        border, thr = int(rng.integers(3, 24)), int(rng.choice([0, 5, 20, 60, 255]))
        tag = (t, w, h, vstep, border, thr)
        stale = rng.integers(0, 256, (rows, vstep), dtype=np.uint8)      # `out` may only change where the reference writes
        o_g, o_o = stale.copy(), stale.copy()
        pl.fastDetect(w, h, img, o_g, thr, border=border, ctx=gpu_ctx)
        orc.fast_detect(img, o_o, w, h, thr, border=border)
        assert (o_g == o_o).all(), tag
        if border >= 4:
            hthr = int(rng.choice([-(1 << 31), 0, 1 << 15, 1 << 24]))
            z_g = np.zeros((rows, vstep), np.uint8)
            z_o = z_g.copy()
            pl.fastDetect(w, h, img, z_g, thr, border=border, ctx=gpu_ctx)
            orc.fast_detect(img, z_o, w, h, thr, border=border)
            pl.fastScoreHarris(w, h, img, hthr, z_g, border=border, ctx=gpu_ctx)
            orc.fast_score_harris(img, z_o, w, h, hthr, border=border)
            assert (z_g == z_o).all(), tag
            lbs, lim = int(rng.choice([0, 0, 1, 2, 3, 4, 5, 6])), int(rng.integers(1, 8))
            kg = pl.fastExtract(w, h, z_g, border=border, logBucketSize=lbs, bucketLimit=lim, ctx=gpu_ctx)
            ko = orc.fast_extract(z_o, w, h, border=border, log_bucket=lbs, bucket_limit=lim)
            assert len(kg) == len(ko) and (kg == ko).all(), tag + (lbs, lim)
            if border >= 16 and len(ko):
                words = int(rng.choice([1, 2, 4, 8]))
                assert (pl.orbCompute(img, ko, words=words, ctx=gpu_ctx) == orc.orb_compute(img, ko, words=words)).all(), tag
        words, nq, nt = int(rng.choice([1, 2, 4, 8])), int(rng.integers(0, 700)), int(rng.integers(0, 900))
        q = rng.integers(0, 2**32, (nq, words), dtype=np.uint64).astype(np.uint32)
        tr = rng.integers(0, 2**32, (nt, words), dtype=np.uint64).astype(np.uint32)
        if nq and nt and rng.integers(0, 2):
            q[: min(nq, nt) // 2] = tr[: min(nq, nt) // 2]           # exact matches and ties
        for a, b in zip(pl.matchHamming(q, tr, ctx=gpu_ctx), orc.match_hamming(q.reshape(nq, words), tr.reshape(nt, words))):
            assert (a == b).all(), (t, nq, nt, words)
        if rows >= 17:
            gw, gh = int(rng.integers(16, vstep + 1)), int(rng.integers(16, rows + 1))
            a, b = img.copy(), img.copy()
            pl.gaussian5x5(gw, gh, a, a, ctx=gpu_ctx)
            orc.gaussian5x5(b, gw, gh)
            assert (a[:gh, :gw] == b[:gh, :gw]).all(), (t, gw, gh, vstep)


def test_random_packed_layouts_match_the_oracle(gpu_ctx, orc):
    """Levels packed side by side on shelves (arbitrary col0, unaligned -> scalar staging path), both
    pipelines; expectation = the oracle run level by level on a flat view shifted to the level's origin."""
    import torch
    from pislam_amd.frontend import OrbFrontend
    dev = torch.device("cuda:0")
    try:
        for t in range(100):
            rng = np.random.default_rng(47000 + t)
            vstep = int(rng.choice([256, 320, 400, 512, 640]))
            levels, x, y, shelf_h = [], int(rng.integers(0, 20)), 0, 0
            for _ in range(int(rng.integers(2, 7))):
                w, h = int(rng.integers(40, 260)), int(rng.integers(40, 140))
                if x + w > vstep:
                    y += shelf_h + int(rng.integers(0, 5))
                    x, shelf_h = int(rng.integers(0, 20)), 0
                w = min(w, vstep - x)
                levels.append((w, h, y, x))
                shelf_h = max(shelf_h, h)
                x += w + int(rng.integers(16, 40))
            rows = y + shelf_h + int(rng.integers(1, 4))
            batch = int(rng.integers(1, 5))
            base = rng.integers(0, 256, (batch, rows // 4 + 2, vstep // 4 + 2), dtype=np.uint8)
            pyr = np.kron(base, np.ones((4, 4), np.uint8))[:, :rows, :vstep].copy()
            if rng.integers(0, 2):
                pyr = (pyr.astype(np.int32) + rng.integers(-6, 7, pyr.shape)).clip(0, 255).astype(np.uint8)
            lbs, lim, border = int(rng.choice([0, 0, 3, 4])), int(rng.integers(1, 6)), int(rng.integers(16, 20))
            opts = dict(pipeline=int(rng.choice([1, 2, 2])), alias=int(rng.integers(0, 2)), run_len=int(rng.choice([0, 1, 5])),
                        strip_rows=int(rng.choice([0, 16, 22])), sub_batches=1 + int(rng.choice([0, 0, 64])) // 48,
                        orb_in_strip=int(rng.integers(0, 2)), tile_cols=int(rng.choice([0, 64, 128])),
                        strip_rows_max=int(rng.choice([0, 44, 56])))
            for k, v in opts.items():
                gpu_ctx.set_option(k, v)
            fe = OrbFrontend(levels, vstep=vstep, rows=rows, max_keypoints=8192, border=border, log_bucket_size=lbs,
                             bucket_limit=lim, ctx=gpu_ctx)
            kp, desc, counts = fe.alloc_outputs(batch, dev)
            fe(torch.from_numpy(pyr).to(dev), kp, desc, counts)
            torch.cuda.synchronize()
            c = counts.cpu().numpy().view(np.uint32)
            k_ = kp.cpu().numpy().view(np.uint32)
            d_ = desc.cpu().numpy().view(np.uint32)
            for b in range(batch):
                exp = []
                for (w, h, r0, c0) in levels:
                    view = np.ascontiguousarray(pyr[b, r0:].reshape(-1)[c0:])
                    view = np.concatenate([view, np.zeros((-len(view)) % vstep + vstep, np.uint8)]).reshape(-1, vstep)
                    lkp, _, _ = orc.pyramid(view, [(w, h, 0)], border=border, log_bucket=lbs, bucket_limit=lim)
                    exp.append(lkp + np.uint32((c0 << 12) | r0))
                exp = np.concatenate(exp)
                if not (c[b] == len(exp) and (k_[b, :len(exp)] == exp).all()):
                    # say WHAT differs: a rerun of the same call tells a nondeterministic launch from a wrong result
                    bad = np.flatnonzero(k_[b, :min(int(c[b]), len(exp))] != exp[:min(int(c[b]), len(exp))])
                    kp2, desc2, counts2 = fe.alloc_outputs(batch, dev)
                    fe(torch.from_numpy(pyr).to(dev), kp2, desc2, counts2)
                    torch.cuda.synchronize()
                    again = kp2.cpu().numpy().view(np.uint32)
                    raise AssertionError((t, b, levels, opts, "lbs", lbs, "count", int(c[b]), len(exp), "first bad", bad[:8].tolist(),
                                          [hex(int(v)) for v in k_[b, bad[:4]]], [hex(int(v)) for v in exp[bad[:4]]],
                                          "same set", bool((np.sort(k_[b, :len(exp)]) == np.sort(exp)).all()),
                                          "rerun equals first run", bool((again[b] == k_[b]).all()),
                                          "rerun equals oracle", bool((again[b, :len(exp)] == exp).all())))
                assert (d_[b, :len(exp)] == orc.orb_compute(pyr[b], exp)).all(), (t, b, levels, opts)
    finally:
        for k, v in dict(pipeline=0, alias=1, run_len=0, strip_rows=0, sub_batches=1, orb_in_strip=0, tile_cols=0,
                         strip_rows_max=0, bucket_select=1, frame=1).items():
            gpu_ctx.set_option(k, v)


def test_random_pyramid_builds_match_the_oracle(gpu_ctx, orc):
    """pislam_pyramid_build_batch on random frame sizes and 7/8, 13/16 chains (with and without the blur) ==
    the oracle running the reference tests' scalar functions in the same order on a zeroed stacked buffer."""
    import torch
    from pislam_amd.capi import PislamError
    from pislam_amd.frontend import PyramidBuilder
    ran = 0
    for t in range(80):
        rng = np.random.default_rng(59000 + t)
        w0, h0 = int(rng.integers(16, 700)), int(rng.integers(16, 400))
        steps = tuple(int(v) for v in rng.integers(1, 3, int(rng.integers(1, 8))))
        blur = bool(rng.integers(0, 2))
        try:
            pb = PyramidBuilder(w0, h0, steps, blur=blur, ctx=gpu_ctx)
        except PislamError:
            continue                                  # a level shrinks to nothing: rejected by the layout
        ran += 1
        batch = int(rng.integers(1, 4))
        frames = rng.integers(0, 256, (batch, h0, w0), dtype=np.uint8)
        d_pyr = torch.full((batch, pb.rows, pb.vstep), 0x5C, dtype=torch.uint8, device="cuda")    # a dirty buffer
        pb(torch.from_numpy(frames).cuda(), d_pyr)
        torch.cuda.synchronize()
        got = d_pyr.cpu().numpy()
        from test_prep import build_defined_mask
        mask = build_defined_mask(pb, steps)
        assert (got[:, ~mask] == 0x5C).all(), (t, w0, h0, steps)      # nothing outside the defined bytes is touched
        got = np.where(mask[None], got, 0)
        for b in range(batch):
            exp = np.zeros((pb.rows, pb.vstep), np.uint8)
            w, h, r0, _ = pb.levels[0]
            exp[r0:r0 + h, :w] = frames[b]
            if blur:
                orc.gaussian5x5(exp[r0:], w, h)
            for k, st in enumerate(steps):
                w, h, r0, _ = pb.levels[k]
                tmp = exp[r0:].copy()                 # out of place: level k -> level k+1 slot
                (orc.bilinear7_8 if st == 1 else orc.bilinear13_16)(tmp, w, h)
                r1 = pb.levels[k + 1][2]
                N, M = (8, 7) if st == 1 else (16, 13)
                oh, ow = -(-h // N) * M, -(-w // N) * M
                exp[r1:r1 + oh, :ow] = tmp[:oh, :ow]
            assert (got[b] == exp).all(), (t, w0, h0, steps, blur)
    assert ran > 60


def test_binary_patterns_score_maps_match_the_oracle(gpu_ctx, orc):
    """Saturated inputs (0/255 checkerboards, stripes, dots, random bits, a few flipped pixels): the extreme
    gradients (-128, the 16-bit wrap of the row-pair products, Harris.h:164-200) and the densest corner
    sets.  Score map (dump hook) and keypoints must equal the oracle on both LDS layouts."""
    import torch
    from pislam_amd.frontend import OrbFrontend
    dev = torch.device("cuda:0")
    try:
        for t in range(200):
            rng = np.random.default_rng(83000 + t)
            w, h = int(rng.integers(60, 200)), int(rng.integers(60, 160))
            vstep, rows = (w + 15) & ~15, h + 2
            kind = int(rng.integers(0, 5))
            yy, xx = np.mgrid[0:rows, 0:vstep]
            lo_v, hi_v = (0, 255) if rng.integers(0, 2) else (int(rng.integers(0, 3)), int(rng.integers(253, 256)))
            if kind == 0:
                m = rng.integers(0, 2, (rows, vstep))
            elif kind == 1:
                p = int(rng.integers(1, 6))
                m = (yy // p + xx // p) & 1
            elif kind == 2:
                p = int(rng.integers(1, 5))
                m = ((xx // p) & 1) if rng.integers(0, 2) else ((yy // p) & 1)
            elif kind == 3:
                p, q = int(rng.integers(1, 5)), int(rng.integers(1, 5))
                m = ((yy // p) & 1) & ((xx // q) & 1)
            else:
                m = np.kron(rng.integers(0, 2, (rows // 3 + 2, vstep // 3 + 2)), np.ones((3, 3), np.int64))[:rows, :vstep]
            img = np.where(m > 0, hi_v, lo_v).astype(np.uint8)
            if rng.integers(0, 3) == 0:
                flip = rng.integers(0, 20, img.shape) == 0
                img = np.where(flip, rng.integers(0, 256, img.shape), img).astype(np.uint8)
            thr = int(rng.choice([0, 1, 20, 100, 254]))
            hthr = int(rng.choice([-(1 << 31), -(1 << 20), 0, 1 << 15, 1 << 26]))
            gpu_ctx.set_option("pipeline", 2)
            gpu_ctx.set_option("dump_score", 1)
            gpu_ctx.set_option("alias", int(rng.integers(0, 2)))
            fe = OrbFrontend([(w, h, 0)], vstep=vstep, rows=rows, max_keypoints=8192, fast_threshold=thr,
                             harris_threshold=hthr, ctx=gpu_ctx)
            kp, desc, counts = fe.alloc_outputs(1, dev)
            fe(torch.from_numpy(img[None].copy()).to(dev), kp, desc, counts)
            torch.cuda.synchronize()
            okp, _, _, osc = orc.pyramid(img, [(w, h, 0)], fast_threshold=thr, harris_threshold=hthr, return_score=True)
            c = int(counts.cpu().numpy().view(np.uint32)[0])
            m_ = min(len(okp), 8192)
            assert (fe.score_map(0) == osc).all(), (t, w, h, kind, thr, hthr)
            assert c == len(okp) and (kp.cpu().numpy().view(np.uint32)[0, :m_] == okp[:m_]).all(), (t, w, h, kind, thr, hthr)
    finally:
        for k, v in dict(pipeline=0, alias=1, dump_score=0).items():
            gpu_ctx.set_option(k, v)
