"""Host and device cost of one count all-gather through the C ABI on a 1-rank RCCL communicator (GPU box):
host time of dist_fence + dist_allgather_counts per call, and the device time between the event before and after."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pislam_amd import capi
from pislam_amd.capi import Context
dev = torch.device("cuda:0")
s = torch.cuda.Stream(dev)
ctx = Context(device=0, stream=s.cuda_stream)
ctx.set_option("dist_rccl_single", 1)
ctx.dist_init(capi.dist_unique_id(), 0, 1)
counts = torch.zeros(256, dtype=torch.int32, device=dev)
outs = [torch.zeros(256, dtype=torch.int32, device=dev) for _ in range(2)]
for _ in range(20):
    ctx.dist_fence(2); ctx.dist_allgather_counts(counts, outs[0])
ctx.dist_synchronize(); torch.cuda.synchronize()
N = 500
t0 = time.perf_counter()
for i in range(N):
    ctx.dist_fence(2)
    ctx.dist_allgather_counts(counts, outs[i & 1])
t1 = time.perf_counter()
ctx.dist_synchronize(); torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue per exchange: {(t1 - t0) / N * 1e6:.1f} us; with completion: {(t2 - t0) / N * 1e6:.1f} us")
# a small kernel on the launch stream between exchanges (dependency chain like a step)
x = torch.zeros(1024, device=dev)
with torch.cuda.stream(s):
    t0 = time.perf_counter()
    for i in range(N):
        ctx.dist_fence(2)
        x.add_(1.0)
        ctx.dist_allgather_counts(counts, outs[i & 1])
    t1 = time.perf_counter()
    ctx.dist_synchronize(); torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f"with a dependent kernel per step: host {(t1 - t0) / N * 1e6:.1f} us, total {(t2 - t0) / N * 1e6:.1f} us per step")
with torch.cuda.stream(s):
    t0 = time.perf_counter()
    for i in range(N):
        x.add_(1.0)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f"the kernel alone: {(t2 - t0) / N * 1e6:.1f} us per step")
ctx.dist_finalize()
