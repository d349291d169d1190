#!/usr/bin/env python3
"""Probe: do two independent front-end pipelines on two streams (batch k on stream k % 2, each with its own context,
workspace and outputs) overlap usefully — the gather+ORB kernel of one batch under the strip kernel of the next?
   python tools/probes/overlap_probe.py [steps]      (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pislam_amd import synth
from pislam_amd.capi import Context
from pislam_amd.frontend import OrbFrontend

K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = 256
dev = torch.device("cuda:0")
levels = synth.level_table()
host = synth.make_batch(0, 16)
d_pyr = torch.from_numpy(host).to(dev)[torch.arange(B, device=dev) % 16].contiguous()


def make(n):
    pipes = []
    for i in range(n):
        s = torch.cuda.Stream(dev)
        ctx = Context(device=0, stream=s.cuda_stream)
        fe = OrbFrontend(levels, vstep=640, rows=2210, max_keypoints=4096, ctx=ctx)
        fe.reserve(B)
        out = fe.alloc_outputs(B, dev)
        with torch.cuda.stream(s):
            fe(d_pyr, *out)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fe(d_pyr, *out)
        pipes.append((s, ctx, fe, out, g))
    return pipes


def run(pipes, steps):
    for i in range(steps):
        s, _, _, _, g = pipes[i % len(pipes)]
        with torch.cuda.stream(s):
            g.replay()
    torch.cuda.synchronize()


for n in (1, 2, 3):
    pipes = make(n)
    run(pipes, 400)                       # clocks
    t0 = time.perf_counter()
    run(pipes, K)
    dt = time.perf_counter() - t0
    kp = int(pipes[0][3][2].sum().item())
    print(f"{n} stream(s): {dt / K * 1e3:.4f} ms per batch, {kp * K / dt:.4e} kp+desc/s")
    ref = pipes[0][3][2].clone()
    assert all(torch.equal(p[3][2], ref) for p in pipes)
