"""Long-running seeded fuzz campaign of the batch path against the oracle (test infrastructure; run on the GPU box):
    python tests/fuzz_campaign.py --seeds 2000 [--start 100000] [--wide]
The cases are the generators of tests/test_gpu_fuzz.py (narrow levels, every tuning option) or, with --wide, levels
up to 704 columns and 320 rows (several full 256-column prefilter steps plus a packed tail step, strips up to 64 rows,
x-tiles as work items, the longest-first launch order).  Prints every mismatch and exits non-zero on any."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def make_wide(seed):
    rng = np.random.default_rng(seed)
    nl = int(rng.integers(1, 5))
    vstep = int(rng.choice([448, 512, 640, 704]))
    levels, row = [], 0
    for _ in range(nl):
        w, h = int(rng.integers(200, vstep + 1)), int(rng.integers(60, 320))
        levels.append((w, h, row))
        row += h
    rows = row + int(rng.integers(0, 3))
    batch = int(rng.integers(1, 4))
    base = rng.integers(0, 256, (batch, rows // 4 + 2, vstep // 4 + 2), dtype=np.uint8)
    pyr = np.kron(base, np.ones((4, 4), np.uint8))[:, :rows, :vstep].copy()
    amp = int(rng.choice([0, 3, 6, 12]))
    if amp:
        pyr = (pyr.astype(np.int32) + rng.integers(-amp, amp + 1, pyr.shape)).clip(0, 255).astype(np.uint8)
    par = dict(border=int(rng.choice([16, 16, 17, 18, 19, 20, 24])), fast_threshold=int(rng.choice([5, 20, 20, 40])),
               harris_threshold=int(rng.choice([0, 1 << 10, 1 << 15, 1 << 15, 1 << 20])),
               log_bucket_size=int(rng.choice([0, 0, 3, 4, 5])), bucket_limit=int(rng.integers(1, 6)),
               words=int(rng.choice([4, 8])), max_keypoints=int(rng.choice([2000, 16384])))
    opts = dict(pipeline=2, alias=int(rng.choice([1, 1, 0])), run_len=int(rng.choice([0, 1, 2, 3, 5])),
                strip_rows=int(rng.choice([0, 0, 0, 16, 20])), sub_batches=1 + int(rng.choice([0, 0, 0, 128])) // 48,
                orb_in_strip=int(rng.choice([0, 0, 1])), tile_cols=int(rng.choice([0, 0, -1, 192, 256, 320, 448])),
                strip_rows_max=int(rng.choice([0, 28, 36, 44, 56, 64])), run_order=int(rng.integers(0, 2)),
                bucket_select=int(rng.choice([1, 1, 0])))
    opts["frame"] = int(rng.choice([8, 1, 0]))      # (drawn last: earlier campaigns' configurations keep their seeds)
    return levels, vstep, rows, pyr, par, opts


def matcher_campaign(args):
    """pislam_match_hamming on random set sizes / widths / tie structures, matrix-core and VALU kernels, vs the oracle."""
    from oracle import orc
    from pislam_amd import frontend
    from pislam_amd.capi import Context
    ctx = Context(device=0)
    bad, t0 = 0, time.time()
    for seed in range(args.start, args.start + args.seeds):
        rng = np.random.default_rng(seed)
        words = int(rng.choice([1, 2, 4, 8]))
        nq, nt = int(rng.integers(0, 400)), int(rng.integers(0, 400))
        kind = int(rng.integers(0, 4))
        if kind == 0:                                   # random bits
            t = rng.integers(0, 2**32, size=(nt, words), dtype=np.uint64).astype(np.uint32)
            q = rng.integers(0, 2**32, size=(nq, words), dtype=np.uint64).astype(np.uint32)
        elif kind == 1:                                 # few distinct descriptors: many exact ties
            pool = rng.integers(0, 2**32, size=(4, words), dtype=np.uint64).astype(np.uint32)
            t = pool[rng.integers(0, 4, nt)] if nt else np.zeros((0, words), np.uint32)
            q = pool[rng.integers(0, 4, nq)] if nq else np.zeros((0, words), np.uint32)
        elif kind == 2:                                 # sparse / dense bit patterns
            t = (rng.integers(0, 2**32, size=(nt, words), dtype=np.uint64) & rng.integers(0, 2**32, size=(nt, words), dtype=np.uint64)).astype(np.uint32)
            q = (rng.integers(0, 2**32, size=(nq, words), dtype=np.uint64) | rng.integers(0, 2**32, size=(nq, words), dtype=np.uint64)).astype(np.uint32)
        else:                                           # queries = noisy copies of train descriptors
            t = rng.integers(0, 2**32, size=(nt, words), dtype=np.uint64).astype(np.uint32)
            if nt:
                q = t[rng.integers(0, nt, nq)] ^ (np.uint32(1) << rng.integers(0, 32, size=(nq, words)).astype(np.uint32))
            else:
                q = np.zeros((nq, words), np.uint32)
        q, t = np.ascontiguousarray(q).reshape(nq, words), np.ascontiguousarray(t).reshape(nt, words)
        exp = orc.match_hamming(q, t)
        for mf in (1, 0):
            ctx.set_option("match_mfma", mf)
            got = frontend.matchHamming(q, t, ctx=ctx)
            if not all((g == e).all() for g, e in zip(got, exp)):
                bad += 1
                print("MISMATCH", seed, "mfma" if mf else "valu", words, nq, nt, kind, flush=True)
        if (seed - args.start) % 1000 == 999:
            print(f"[{seed - args.start + 1} cases, {bad} bad, {time.time() - t0:.0f} s]", flush=True)
    print(f"{args.seeds} matcher cases from seed {args.start}: {bad} mismatches, {time.time() - t0:.0f} s")
    return 1 if bad else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=500)
    ap.add_argument("--start", type=int, default=100000)
    ap.add_argument("--wide", action="store_true")
    ap.add_argument("--matcher", action="store_true", help="random descriptor sets through both matcher kernels instead")
    args = ap.parse_args()
    if args.matcher:
        return matcher_campaign(args)
    import torch
    from oracle import orc
    from pislam_amd.capi import Context
    from pislam_amd.frontend import OrbFrontend
    from test_gpu_fuzz import make_case
    dev = torch.device("cuda:0")
    ctx = Context(device=0)
    bad, t0 = 0, time.time()
    for seed in range(args.start, args.start + args.seeds):
        levels, vstep, rows, pyr, par, opts = (make_wide if args.wide else make_case)(seed)
        for k, v in opts.items():
            ctx.set_option(k, v)
        fe = OrbFrontend(levels, vstep=vstep, rows=rows, ctx=ctx, **par)
        kp, desc, counts = fe.alloc_outputs(len(pyr), dev)
        fe(torch.from_numpy(pyr).to(dev), kp, desc, counts)
        torch.cuda.synchronize()
        c = counts.cpu().numpy().view(np.uint32)
        k = kp.cpu().numpy().view(np.uint32)
        d = desc.cpu().numpy().view(np.uint32)
        for b in range(len(pyr)):
            okp, odesc, _ = orc.pyramid(pyr[b], levels, fast_threshold=par["fast_threshold"],
                                        harris_threshold=par["harris_threshold"], border=par["border"],
                                        log_bucket=par["log_bucket_size"], bucket_limit=par["bucket_limit"], words=par["words"])
            m = min(len(okp), par["max_keypoints"])
            if c[b] != len(okp) or not (k[b, :m] == okp[:m]).all() or not (d[b, :m].reshape(m, par["words"]) == odesc[:m]).all():
                bad += 1
                print("MISMATCH", seed, b, par, opts, levels, "count", int(c[b]), len(okp), flush=True)
                break
        if (seed - args.start) % 200 == 199:
            print(f"[{seed - args.start + 1} cases, {bad} bad, {time.time() - t0:.0f} s]", flush=True)
    print(f"{args.seeds} cases from seed {args.start}{' (wide)' if args.wide else ''}: {bad} mismatches, {time.time() - t0:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
