"""CPU suite: bench.py's N>1 supervision — every torchrun rank is a supervisor that starts its worker under a
timeout and, on a failure or a hang of ANY rank's worker, walks a ladder of safer configurations together with
the other supervisors (bench.py LADDER).  Failures are injected per attempt and stage (PISLAM_BENCH_INJECT) in
the CPU self-test mode (fake counts, gloo) — the supervision code is the one the 8-GPU run uses."""
import json
import os
import subprocess
import sys
import time

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")
CLEAN = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PISLAM_BENCH_INJECT", "PISLAM_BENCH_SPIN_SKEW")


def run_bench(extra, inject=None, timeout=300, skew=None):
    env = {k: v for k, v in os.environ.items() if k not in CLEAN and not k.startswith("TORCHELASTIC_")}
    if inject:
        env["PISLAM_BENCH_INJECT"] = inject
    if skew:
        env["PISLAM_BENCH_SPIN_SKEW"] = skew
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--selftest-spawn"] + extra, capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None), time.monotonic() - t0


def test_clean_run_drops_nothing():
    r, d, _ = run_bench([])
    assert r.returncode == 0, r.stderr[-2000:]
    assert d["n_gpus"] == 2 and d["exchange_ok"] is True
    assert d["config"]["dist_fallbacks"] == [] and d["config"]["attempt"] == 0 and "rccl_ranks" in d["config"]
    assert len([ln for ln in r.stdout.splitlines() if ln.startswith("{")]) == 1      # ONE JSON line


@pytest.mark.parametrize("skew", ["1:3", "0:4"])
def test_ranks_whose_clocks_disagree_about_the_ramp_issue_the_same_collectives(skew):
    """The clock-ramp before the timed region is time-based and every step in it issues a count all-gather: a rank
    whose own clock ends the ramp one group early would leave its peers with collectives that never complete
    (round-3 bug: 8*S all-gathers of difference, a 300 s attempt lost).  One rank spins 3x / 4x longer here; the run
    must complete on attempt 0, drop nothing, and every rank must report the same number of all-gathers."""
    r, d, dt = run_bench(["--spin-s", "0.4", "--selfcheck-timeout", "10", "--attempt-timeout", "60"], skew=skew)
    assert r.returncode == 0, r.stderr[-3000:]
    c = d["config"]
    assert c["attempt"] == 0 and c["dist_fallbacks"] == [] and d["exchange_ok"] is True
    per_rank = c["count_allgathers_per_rank"]
    assert len(per_rank) == 2 and per_rank[0] == per_rank[1]
    # the ramp lasted as long as the SLOWEST rank wanted (>= 1.2 s of 2 ms groups), not as long as rank 0's clock said
    assert c["ramp_groups"] >= 20 and per_rank[0] == 2 + 8 * c["ramp_groups"] + 1
    assert "next rung" not in r.stderr


def test_clock_ramp_runs_an_agreed_number_of_groups():
    """Unit test of bench.clock_ramp: the continue decision comes from agree_any, not from the local clock."""
    import bench
    calls, votes = [], []

    def agree(flag):                     # a peer that wants exactly 5 groups, whatever this rank's clock says
        votes.append(flag)
        return len(votes) <= 5

    assert bench.clock_ramp(0.0, lambda: calls.append(1), agree) == 5
    assert len(calls) == 5 and votes[0] is False
    assert bench._spin_seconds(0.5, 1) == 0.5


@pytest.mark.parametrize("inject, rungs, final", [
    # a worker that raises at start-up on rank 1 only: every rank moves to the next rung together
    ("0:start:fail:1", 1, {"graph": 0}),
    # a hang inside the start-up self-check (what a wedged collective looks like), then a crash one rung later
    ("0:selfcheck:hang:0,1:run:exit:1", 2, {"graph": 0, "streams": 1}),
    # failures at every stage, one per rung, down to the last resort
    ("0:start:exit,1:selfcheck:fail:1,2:run:fail:0,3:selfcheck:hang:1", 4, {"graph": 0, "streams": 1, "exchange": "torch",
                                                                             "dist_backend": "gloo"}),
])
def test_failures_walk_the_ladder_in_order(inject, rungs, final):
    from bench import LADDER
    r, d, _ = run_bench(["--selfcheck-timeout", "4", "--attempt-timeout", "40"], inject=inject)
    assert r.returncode == 0, r.stderr[-3000:]
    assert d["n_gpus"] == 2 and d["exchange_ok"] is True
    want = [label for label, _ in LADDER[1:1 + rungs]]
    assert d["config"]["dist_fallbacks"] == want, d["config"]
    assert d["config"]["attempt"] == rungs
    for k, v in final.items():
        assert d["config"][k] == v, (k, d["config"])
    assert r.stderr.count("-> next rung") == rungs


def test_a_worker_that_never_returns_is_killed_by_its_supervisor():
    """`hang` at the run stage is past the worker's own self-check watchdog: the supervisor's attempt timeout ends it."""
    r, d, dt = run_bench(["--selfcheck-timeout", "3", "--attempt-timeout", "6"], inject="0:run:hang:1")
    assert r.returncode == 0, r.stderr[-3000:]
    assert d["config"]["attempt"] == 1 and "killed" in r.stderr
    assert dt < 120


def test_wall_clock_cap_ends_the_run_with_an_error():
    """Nothing works: the run must END (rc != 0, a message naming what was tried), not hang."""
    inject = ",".join(f"{k}:start:fail" for k in range(8))
    r, d, dt = run_bench(["--selfcheck-timeout", "3", "--attempt-timeout", "20", "--wall-cap", "200"], inject=inject)
    assert r.returncode != 0 and d is None
    assert "no configuration completed" in r.stderr and "attempt 4" in r.stderr
    assert dt < 200


def test_rungs_that_drop_nothing_are_skipped():
    """--graph 0 --streams 1 already: the first two rungs would change nothing and are not retried."""
    r, d, _ = run_bench(["--graph", "0", "--streams", "1", "--selfcheck-timeout", "4", "--attempt-timeout", "40"],
                        inject="0:start:fail")
    assert r.returncode == 0, r.stderr[-3000:]
    assert d["config"]["attempt"] == 1 and d["config"]["exchange"] == "torch"
    assert len(d["config"]["dist_fallbacks"]) == 1 and "torch.distributed" in d["config"]["dist_fallbacks"][0]


def test_effective_cores_reads_the_cgroup_quota(tmp_path, monkeypatch):
    import bench
    vis, eff, how = bench.effective_cores()
    assert vis >= 1 and 0 < eff <= vis and isinstance(how, str)
