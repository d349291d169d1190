"""One process per GPU: shard a batch of independent pyramids, all-gather the keypoint counts.

The ORB front-end has no cross-pyramid state, so the data path needs NO collective:
rank r of G owns the contiguous pyramid range [r*B/G, (r+1)*B/G) (here: a fixed per-rank batch, weak
scaling).  The only exchange is one all-gather of the per-pyramid keypoint counts (B/G uint32 per
rank) so that every rank knows the global offsets/total — RCCL over xGMI on GPUs ("nccl" backend),
gloo on CPU for the tests.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend: str | None = None):
    """Initialise torch.distributed from the torchrun environment (no-op for WORLD_SIZE=1)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def shard_range(global_batch: int, rank: int, world: int):
    """Contiguous split; the first (global_batch % world) ranks take one extra pyramid."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_counts(local_counts: torch.Tensor, world: int, shard_sizes=None) -> torch.Tensor:
    """All-gather per-pyramid keypoint counts -> int32 [global_batch] in global pyramid order.

    Equal shards use one all_gather_into_tensor (a single small RCCL collective); ragged shards pad
    to the largest shard and trim."""
    if world == 1:
        return local_counts
    if dist.get_backend() == "gloo" and local_counts.is_cuda:     # test mode: gloo moves host tensors
        return gather_counts(local_counts.cpu(), world, shard_sizes).to(local_counts.device)
    n = local_counts.numel()
    if shard_sizes is None or len(set(shard_sizes)) == 1:
        out = torch.empty(n * world, dtype=local_counts.dtype, device=local_counts.device)
        dist.all_gather_into_tensor(out, local_counts.contiguous())
        return out
    m = max(shard_sizes)
    padded = torch.zeros(m, dtype=local_counts.dtype, device=local_counts.device)
    padded[:n] = local_counts
    out = torch.empty(m * world, dtype=local_counts.dtype, device=local_counts.device)
    dist.all_gather_into_tensor(out, padded)
    return torch.cat([out[r * m:r * m + shard_sizes[r]] for r in range(world)])


class CountExchange:
    """The per-step count all-gather taken OFF the critical path: step i's counts are gathered on the
    collective's own stream while step i+1 computes into the other of two output sets (the all-gather is
    latency-bound — 1 KiB per rank — and would otherwise add its ~tens of microseconds to every ~0.4 ms
    step).  `start(counts)` after the step's kernels are enqueued; `finish()` returns the last gathered
    tensor.  The caller alternates output sets, so counts[i] stays valid until step i+2, which first
    waits for its all-gather."""

    def __init__(self, world: int, always_collective: bool = False):
        self.world = world
        self.always = always_collective      # tests: run the collective path on a 1-rank group too
        self.pending = [None, None]          # (work, out) per output set
        self.last = None
        self.i = 0

    def start(self, local_counts: torch.Tensor):
        slot = self.i & 1
        self.i += 1
        if self.world == 1 and not self.always:
            self.last = local_counts
            return
        if dist.get_backend() == "gloo":     # test mode (host tensors): synchronous
            self.last = gather_counts(local_counts, self.world)
            return
        prev = self.pending[slot]
        if prev is not None:
            prev[0].wait()                   # two steps old: done long ago; orders the buffer reuse
        out = torch.empty(local_counts.numel() * self.world, dtype=local_counts.dtype, device=local_counts.device)
        work = dist.all_gather_into_tensor(out, local_counts, async_op=True)
        self.pending[slot] = (work, out)
        self.last = (work, out)

    def finish(self) -> torch.Tensor:
        for p in self.pending:
            if p is not None:
                p[0].wait()
        self.pending = [None, None]
        if isinstance(self.last, tuple):
            return self.last[1]
        return self.last


def global_offsets(all_counts: torch.Tensor) -> torch.Tensor:
    """Exclusive prefix sum: where pyramid i's keypoints start in a global concatenation."""
    c = all_counts.to(torch.int64)
    return torch.cumsum(c, 0) - c
