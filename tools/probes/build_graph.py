"""probe: pyramid build of 64 720p frames — eager launches vs one hipGraph replay (launch floors between dependent kernels)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pislam_amd import synth
from pislam_amd.capi import Context
from pislam_amd.frontend import PyramidBuilder
dev = torch.device("cuda:0")
s = torch.cuda.Stream(dev)
ctx = Context(device=0, stream=s.cuda_stream)
B = 64
fr = torch.from_numpy(synth.make_many(range(16), workers=8, kind="level0", w0=1280, h0=720)).to(dev)
fr = fr[torch.arange(B, device=dev) % 16].contiguous()
pb = PyramidBuilder(1280, 720, ctx=ctx)
pyr = torch.zeros((B, pb.rows, pb.vstep), dtype=torch.uint8, device=dev)
with torch.cuda.stream(s):
    pb(fr, pyr)
torch.cuda.synchronize()
def timeit(fn, n=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s):
        for _ in range(5): fn()
        e0.record(s)
        for _ in range(n): fn()
        e1.record(s)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
eager = timeit(lambda: pb(fr, pyr, margins_clean=True))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    pb(fr, pyr, margins_clean=True)
graph = timeit(lambda: g.replay())
print(f"build of {B} 720p frames: eager {eager*1e3:.1f} us, hipGraph replay {graph*1e3:.1f} us")
