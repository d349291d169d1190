"""CPU suite: the N>1 path (shard ranges + count all-gather) with world_size 2 on gloo."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch
    import torch.distributed as dist
    from pislam_amd import dist as pdist
    rank, local_rank, world = pdist.init(backend="gloo")
    assert world == 2 and dist.get_backend() == "gloo"
    mode = sys.argv[1]
    if mode == "equal":
        # weak scaling shape of bench.py: every rank owns B pyramids
        B = 5
        local = torch.arange(B, dtype=torch.int32) + 100 * rank
        allc = pdist.gather_counts(local, world)
        exp = torch.cat([torch.arange(B, dtype=torch.int32) + 100 * r for r in range(world)])
        assert torch.equal(allc, exp), (allc, exp)
        off = pdist.global_offsets(allc)
        assert off[0] == 0 and off[-1] == int(exp[:-1].sum())
    elif mode == "exchange":
        # bench.py's per-step exchange: two alternating count buffers, the last step's gather is returned
        x = pdist.CountExchange(world)
        bufs = [torch.zeros(4, dtype=torch.int32) for _ in range(2)]
        for i in range(5):
            x.before_step()                  # orders the reuse of bufs[i & 1] against its previous all-gather
            bufs[i & 1].fill_(10 * i + rank)
            x.start(bufs[i & 1])
        allc = x.finish()
        assert allc.tolist() == [40] * 4 + [41] * 4, allc
    else:
        # strong scaling / ragged: 7 pyramids over 2 ranks -> 4 + 3, results independent of the sharding
        G = 7
        lo, hi = pdist.shard_range(G, rank, world)
        sizes = [pdist.shard_range(G, r, world)[1] - pdist.shard_range(G, r, world)[0] for r in range(world)]
        assert sizes == [4, 3] and sum(sizes) == G
        counts_global = torch.tensor([11, 0, 5, 7, 3, 2, 9], dtype=torch.int32)
        allc = pdist.gather_counts(counts_global[lo:hi].clone(), world, shard_sizes=sizes)
        assert torch.equal(allc, counts_global), allc
        assert pdist.global_offsets(allc).tolist() == [0, 11, 11, 16, 23, 26, 28]
    dist.barrier()
    dist.destroy_process_group()
    print("ok", rank)
''') % ROOT


@pytest.mark.parametrize("mode", ["equal", "ragged", "exchange"])
def test_two_rank_gloo_count_allgather(tmp_path, mode):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), mode], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=120)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "ok" in o


def test_shard_ranges_cover_batch_exactly():
    from pislam_amd import dist as pdist
    for G in (1, 7, 8, 256, 2048, 2049):
        for world in (1, 2, 3, 8):
            r = [pdist.shard_range(G, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == G
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_c_abi_shard_equals_python_shard():
    """pislam_dist_shard (the C++ host's split, include/pislam_hip.h) == pislam_amd.dist.shard_range."""
    from pislam_amd import capi, dist as pdist
    for G in (0, 1, 7, 8, 256, 2048, 2049):
        for world in (1, 2, 3, 8):
            for r in range(world):
                assert capi.dist_shard(G, r, world) == pdist.shard_range(G, r, world)
    with pytest.raises(capi.PislamError):
        capi.dist_shard(8, 2, 2)


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` WITHOUT a torchrun environment must start two ranks itself (the driver's plain
    invocation) — here in the CPU self-test mode (launch + rendezvous + count exchange)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-spawn"],
                       capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["exchange_ok"] is True
    # a world size that contradicts --gpus is refused, never silently benchmarked as something else
    env2 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-spawn"],
                       capture_output=True, text=True, timeout=120, env=env2, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
