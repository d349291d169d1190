"""usage (GPU box): python tools/probes/chain_probe.py — build time of 64 720p frames with parts of pp::k_bilinear_chain's
protocol switched off (test bits: results invalid, timing only)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pislam_amd.capi import Context
from pislam_amd.frontend import PyramidBuilder
ctx = Context(device=0)
pb = PyramidBuilder(1280, 720, ctx=ctx)
B = 64
fr = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (B, 720, 1280), dtype=np.uint8)).cuda()
pyr = torch.zeros((B, pb.rows, pb.vstep), dtype=torch.uint8, device="cuda")
def t(label, chain, test):
    ctx.set_option("build_chain", chain); ctx.set_option("frame_test", test)
    for _ in range(3): pb(fr, pyr, margins_clean=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): pb(fr, pyr, margins_clean=True)
    e1.record(); torch.cuda.synchronize()
    print(label, round(e0.elapsed_time(e1) / 20 * 1e3, 1), "us per build", flush=True)
    try:
        ctx.synchronize()
    except Exception as e:
        print("   (fault reported:", str(e)[:60], ")"); ctx.set_option("frame_rearm", 1)
t("per-level launches", 0, 0)
t("chain", 1, 0)
t("chain, no home check", 1, 32)
t("chain, no polls", 1, 4 | 32)
t("chain, no polls, no band adds", 1, 4 | 8 | 32)
t("chain, no polls, no band adds, no done counter", 1, 4 | 8 | 16 | 32)
t("chain, polls but no done counter", 1, 16 | 32)
