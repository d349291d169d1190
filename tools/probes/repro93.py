import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from oracle import orc
from pislam_amd.frontend import OrbFrontend, default_context
ctx = default_context()
dev = torch.device("cuda:0")
def case(t):
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
    return levels, vstep, rows, batch, pyr, lbs, lim, border, opts
def run(t, over=None):
    levels, vstep, rows, batch, pyr, lbs, lim, border, opts = case(t)
    opts = dict(opts, **(over or {}))
    for k, v in opts.items(): ctx.set_option(k, v)
    fe = OrbFrontend(levels, vstep=vstep, rows=rows, max_keypoints=8192, border=border, log_bucket_size=lbs, bucket_limit=lim, ctx=ctx)
    kp, desc, counts = fe.alloc_outputs(batch, dev)
    fe(torch.from_numpy(pyr).to(dev), kp, desc, counts); torch.cuda.synchronize()
    c = counts.cpu().numpy().view(np.uint32); k_ = kp.cpu().numpy().view(np.uint32)
    bad = 0
    for b in range(batch):
        exp = []
        for (w, h, r0, c0) in levels:
            view = np.ascontiguousarray(pyr[b, r0:].reshape(-1)[c0:])
            view = np.concatenate([view, np.zeros((-len(view)) % vstep + vstep, np.uint8)]).reshape(-1, vstep)
            lkp, _, _ = orc.pyramid(view, [(w, h, 0)], border=border, log_bucket=lbs, bucket_limit=lim)
            exp.append(lkp + np.uint32((c0 << 12) | r0))
        exp = np.concatenate(exp)
        ok = c[b] == len(exp) and (k_[b, :len(exp)] == exp).all()
        if not ok:
            bad += 1
            idx = np.flatnonzero(k_[b, :len(exp)] != exp)
            print("  b", b, "count", int(c[b]), len(exp), "first bad", idx[:6], [hex(int(v)) for v in k_[b, idx[:4]]], [hex(int(v)) for v in exp[idx[:4]]], "sorted-equal", (np.sort(k_[b,:len(exp)])==np.sort(exp)).all())
    print("case", t, "lbs", lbs, "lim", lim, opts, "bad", bad, flush=True)
import itertools
nbad = 0
for rep in range(400):
    for t in (91, 93, 94, 96, 89):
        levels, vstep, rows, batch, pyr, lbs, lim, border, opts = case(t)
        for k, v in opts.items(): ctx.set_option(k, v)
        fe = OrbFrontend(levels, vstep=vstep, rows=rows, max_keypoints=8192, border=border, log_bucket_size=lbs, bucket_limit=lim, ctx=ctx)
        kp, desc, counts = fe.alloc_outputs(batch, dev)
        fe(torch.from_numpy(pyr).to(dev), kp, desc, counts); torch.cuda.synchronize()
        c = counts.cpu().numpy().view(np.uint32); k_ = kp.cpu().numpy().view(np.uint32)
        if rep == 0:
            REF = globals().setdefault("REF", {})
            REF[t] = (c.copy(), k_.copy())
        else:
            rc, rk = REF[t]
            if not (c == rc).all() or any(not (k_[b, :c[b]] == rk[b, :c[b]]).all() for b in range(batch)):
                nbad += 1
                for b in range(batch):
                    idx = np.flatnonzero(k_[b, :c[b]] != rk[b, :c[b]])
                    if len(idx):
                        print("rep", rep, "case", t, "b", b, "counts", int(c[b]), int(rc[b]), "nbad idx", len(idx), idx[:8], [hex(int(v)) for v in k_[b, idx[:4]]], [hex(int(v)) for v in rk[b, idx[:4]]], "sorted-equal", (np.sort(k_[b,:c[b]])==np.sort(rk[b,:c[b]])).all(), flush=True)
print("total nondeterministic results:", nbad)
