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
        return local_counts.clone()
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


def global_offsets(all_counts: torch.Tensor) -> torch.Tensor:
    """Exclusive prefix sum: where pyramid i's keypoints start in a global concatenation."""
    c = all_counts.to(torch.int64)
    return torch.cumsum(c, 0) - c
