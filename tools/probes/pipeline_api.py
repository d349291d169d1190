#!/usr/bin/env python3
"""usage: python tools/probes/pipeline_api.py [depth ...]   (GPU box)
ms per batch of 256 VGA pyramids through pislam_pipeline_* (eager launches, batches in flight on `depth` lanes)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from pislam_amd import capi, synth
from pislam_amd.frontend import OrbFrontend

dev = torch.device("cuda:0")
levels = synth.level_table()
B = 256
pyr = torch.from_numpy(synth.make_batch(0, 32)).to(dev)[torch.arange(B, device=dev) % 32].contiguous()
fe = OrbFrontend(levels, vstep=640, rows=2210, max_keypoints=4096)
for depth in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]:
    pipe = capi.Pipeline(device=0, depth=depth)
    pipe.reserve(fe.params, fe.levels, B)
    outs = [fe.alloc_outputs(B, dev) for _ in range(depth)]
    for k in range(600):
        pipe.submit(fe.params, fe.levels, pyr, *outs[k % depth])
    pipe.synchronize()
    t0 = time.perf_counter()
    K = 200
    for k in range(K):
        pipe.submit(fe.params, fe.levels, pyr, *outs[k % depth])
    pipe.synchronize()
    dt = time.perf_counter() - t0
    kp = int(torch.clamp(outs[0][2], max=4096).sum().item())
    print(f"depth {depth}: {dt / K * 1e3:.4f} ms per batch, {kp * K / dt:.4e} kp+desc/s", flush=True)
    pipe.close()
