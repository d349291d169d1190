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
