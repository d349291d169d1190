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
                strip_rows=int(rng.choice([0, 0, 10, 16, 22, 32])), xtile_cols=int(rng.choice([0, 0, 0, 64, 96])))
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
        for k, v in dict(pipeline=0, alias=1, run_len=0, strip_rows=0, xtile_cols=-1).items():
            gpu_ctx.set_option(k, v)
