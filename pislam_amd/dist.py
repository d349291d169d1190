"""One process per GPU: shard a batch of independent pyramids, all-gather the keypoint counts.

The ORB front-end has no cross-pyramid state (the reference is a per-frame loop,
demo.cpp:77-101), so the data path needs NO collective: rank r of G owns the contiguous pyramid range
[r*B/G, (r+1)*B/G) (here: a fixed per-rank batch, weak scaling).  The only exchange is one all-gather of
the per-pyramid keypoint counts (B/G uint32 per rank) so that every rank knows the global offsets/total.

Data plane on GPUs: the C ABI's `pislam_dist_*` entry points (include/pislam_hip.h) — an RCCL communicator
created with ncclCommInitRank from a unique id, ncclAllGather on the context's collective stream.  The same
entry points serve a C++ host (tools/pislam_demo.cpp).  torch.distributed is only the
control plane here: rendezvous, handing the unique id to every rank, barriers and the max-over-ranks of the
timed region.  On CPU-only ranks, or when several test ranks share one GPU (RCCL refuses two ranks on one
device), the exchange falls back to torch.distributed's own all-gather (gloo) — test mode only.
"""
from __future__ import annotations

import os
import sys

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend: str | None = None):
    """Initialise torch.distributed (the control plane) from the torchrun environment (no-op for WORLD_SIZE=1)."""
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
    """Contiguous split; the first (global_batch % world) ranks take one extra pyramid
    (the same arithmetic as the C ABI's pislam_dist_shard)."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_rccl(ctx, rank: int, world: int, device: torch.device | None = None) -> str | None:
    """Create the context's RCCL communicator through the C ABI (pislam_dist_init): rank 0 draws the
    unique id (pislam_dist_get_unique_id) and the control plane broadcasts it.  Returns None on success or
    the reason the C-ABI path is unavailable — agreed on by ALL ranks, so that they take the same path."""
    from . import capi
    if world == 1:
        ctx.dist_init(None, 0, 1)
        return None
    err = None
    uid = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        try:
            uid = torch.frombuffer(bytearray(capi.dist_unique_id()), dtype=torch.uint8).clone()
        except Exception as e:                      # noqa: BLE001
            err = repr(e)
    on_gpu = dist.get_backend() == "nccl"
    t = uid.to(device) if on_gpu else uid
    flag = torch.tensor([1 if err else 0], dtype=torch.int32, device=device if on_gpu else None)
    dist.broadcast(t, src=0)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if int(flag.item()):
        return err or "rank 0 could not draw an RCCL unique id"
    try:
        ctx.dist_init(bytes(t.cpu().numpy().tobytes()), rank, world)
    except Exception as e:                          # noqa: BLE001
        err = repr(e)
    flag = torch.tensor([1 if err else 0], dtype=torch.int32, device=device if on_gpu else None)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if int(flag.item()):
        try:
            ctx.dist_finalize()
        except Exception:                           # noqa: BLE001
            pass
        return err or "pislam_dist_init failed on another rank"
    return None


def gather_counts(local_counts: torch.Tensor, world: int, shard_sizes=None) -> torch.Tensor:
    """torch.distributed all-gather of per-pyramid keypoint counts -> [global_batch] in global pyramid order
    (control-plane / test path; the GPU data plane is CountExchange over the C ABI).

    Equal shards use one all_gather_into_tensor; ragged shards pad to the largest shard and trim."""
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


class ExchangeHub:
    """ONE communicator and ONE collective stream per process (a capi.Context on which init_rccl succeeded),
    shared by every pipeline of the process: all collectives are issued through it, in host order — the same
    order on every rank (several communicators whose collectives a device may run in different orders on
    different ranks are the documented NCCL/RCCL deadlock pattern)."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.issued = 0                 # collectives issued through this hub so far


class CountExchange:
    """The per-step count all-gather taken OFF the critical path: step i's counts are gathered on the
    collective stream while step i+1 computes into the other of two output sets (the all-gather is
    latency-bound — 1 KiB per rank — and would otherwise add its tens of microseconds to every ~0.3 ms step).

        xchg.before_step()        # orders the reuse of the output set written `sets` steps ago
        <enqueue step i's kernels into output set i % sets>
        xchg.start(counts_i)      # all-gather of step i's counts, asynchronous
        ...
        all_counts = xchg.finish()

    `ctx` (a capi.Context on which init_rccl succeeded) or `hub` (an ExchangeHub: several pipelines sharing one
    communicator) selects the C-ABI RCCL path; `stream` (raw hipStream_t) is the stream this pipeline's kernels
    run on when it is not the hub context's own.  Without ctx / hub the exchange uses torch.distributed (gloo
    test mode, or the fallback bench.py reports as such)."""

    def __init__(self, world: int, ctx=None, always_collective: bool = False, sets: int = 2, hub: ExchangeHub | None = None,
                 stream: int | None = None):
        self.world = world
        self.sets = sets                     # output sets the caller alternates between on this pipeline (1 or 2)
        self.hub = hub if hub is not None else (ExchangeHub(ctx) if ctx is not None else None)
        self.ctx = self.hub.ctx if self.hub is not None else None
        self.stream = stream
        self.always = always_collective      # tests: run the collective path on a 1-rank group too
        self.pending = [None] * sets         # torch path: (work, out) per output set
        self.outs = [None] * sets            # C-ABI path: gathered counts per output set
        self.ticket = [None] * sets          # C-ABI path: hub.issued when the set's last all-gather was issued
        self.last = None
        self.i = 0
        self.collectives = 0                 # all-gathers this exchange has issued (any path): must stay equal across ranks

    @property
    def path(self) -> str:
        if self.world == 1 and not self.always:
            return "none (single GPU)"
        return "pislam_dist_allgather_counts (C ABI, RCCL)" if self.ctx is not None else f"torch.distributed ({dist.get_backend()})"

    def before_step(self):
        """Call before enqueueing a step: the step overwrites the counts buffer whose all-gather was started
        `sets` steps ago (on this pipeline), so the launch stream first waits (on the device) for that collective."""
        if self.world == 1 and not self.always:
            return
        slot = self.i % self.sets
        if self.ctx is not None:
            if self.ticket[slot] is not None:
                self.ctx.dist_fence(self.hub.issued - self.ticket[slot], self.stream)
            return
        prev = self.pending[slot]
        if prev is not None:
            prev[0].wait()
            self.pending[slot] = None

    def start(self, local_counts: torch.Tensor):
        slot = self.i % self.sets
        self.i += 1
        if self.world == 1 and not self.always:
            self.last = local_counts
            return
        self.collectives += 1
        if self.ctx is not None:
            n = local_counts.numel()
            if self.outs[slot] is None or self.outs[slot].numel() != n * self.world:
                self.outs[slot] = torch.empty(n * self.world, dtype=local_counts.dtype, device=local_counts.device)
            self.ctx.dist_allgather_counts(local_counts, self.outs[slot], self.stream)
            self.ticket[slot] = self.hub.issued
            self.hub.issued += 1
            self.last = self.outs[slot]
            return
        if dist.get_backend() == "gloo":     # test mode (host tensors): synchronous
            self.last = gather_counts(local_counts, self.world)
            return
        out = torch.empty(local_counts.numel() * self.world, dtype=local_counts.dtype, device=local_counts.device)
        work = dist.all_gather_into_tensor(out, local_counts, async_op=True)
        self.pending[slot] = (work, out)
        self.last = out

    def finish(self) -> torch.Tensor:
        if self.ctx is not None and not (self.world == 1 and not self.always):
            self.ctx.dist_synchronize()
        for k, p in enumerate(self.pending):
            if p is not None:
                p[0].wait()
                self.pending[k] = None
        return self.last


def collectives_agree(exchanges, world: int, device=None):
    """Every rank must have issued the same number of count all-gathers (a rank that issued more leaves collectives
    that never complete; one that issued fewer blocks its peers).  One all-gather of the totals on the control plane,
    OUTSIDE any timed region.  Returns (ok, per-rank totals)."""
    mine = sum(x.collectives for x in exchanges)
    if world == 1 or not dist.is_initialized():
        return True, [mine]
    on_gpu = dist.get_backend() == "nccl"
    t = torch.tensor([mine], dtype=torch.int64, device=device if on_gpu else None)
    allt = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(allt, t)
    tot = [int(x.item()) for x in allt]
    return len(set(tot)) == 1, tot


def global_offsets(all_counts: torch.Tensor) -> torch.Tensor:
    """Exclusive prefix sum: where pyramid i's keypoints start in a global concatenation."""
    c = all_counts.to(torch.int64)
    return torch.cumsum(c, 0) - c


def self_launch(argv, gpus: int, extra_env=None) -> int:
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks ourselves
    (python -m torch.distributed.run --nnodes=1 --nproc-per-node N on 127.0.0.1 with a free port) and
    pass rank 0's output through.  Returns the launcher's exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "1")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port)] + list(argv)
    return subprocess.call(cmd, env=env)
