#!/usr/bin/env python3
"""bench.py — ORB keypoints+descriptors/sec on synthetic 640x480 8-level x1.2 pyramids.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the whole hot path (fastDetect -> fastScoreHarris -> fastExtract ->
orbCompute on every level of every pyramid) over one device-resident batch of 256 synthetic
pyramids per GPU (BASELINE.json configs[1]); for N>1 each rank owns its own 256 pyramids (weak
scaling, configs[2]) and the step ends with the RCCL all-gather of the per-pyramid keypoint counts.
Prints ONE JSON line on rank 0.

Process structure for N > 1 (the run must not be lost to a hang): every torchrun rank is a SUPERVISOR that
never touches the GPU.  The supervisors (a gloo group of their own) start one WORKER child per rank with a
rendezvous port of its own, wait for it under a timeout, agree on the outcome, and on a failure or a hang kill
the workers and retry one rung further down a ladder of safer configurations (eager launches instead of
hipGraph replay -> one stream -> torch.distributed's all-gather instead of the C-ABI communicator -> counts
gathered on the host through gloo); the JSON line lists what was dropped (`config.dist_fallbacks`).  A
wall-clock cap ends the whole run with rc != 0 and a message if nothing works.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_CYCLES_PER_WAVE_INST = 4.0 # what roofline.valu.issue_frac assumes (see its note)
VALU_SUSTAINED_PER_S = 533e9    # tools/probes/valu_rate.hip on MI355X: 528-538 G wave64 integer instructions/s chip-wide
METRIC = "ORB keypoints+descriptors/sec, 640x480 8-level pyramid"
PROFILE_TAG = "r06"             # profiles/<tag>_counters_<workload>.json holds the PMC passes bench lines quote


# ---------------------------------------------------------------------------------------------------------
# host facts
# ---------------------------------------------------------------------------------------------------------
def effective_cores():
    """CPUs this process can really use: the affinity mask, narrowed by a cgroup CPU quota (cpu.max, v2; or
    cfs_quota_us / cfs_period_us, v1).  Returns (visible, effective, how)."""
    vis = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    eff, how = float(vis), "sched_getaffinity"
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            lim = float(q) / float(p)
            if lim < eff:
                eff, how = lim, f"cgroup cpu.max {q}/{p}"
    except Exception:                                   # noqa: BLE001
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and q / p < eff:
                eff, how = q / p, f"cgroup cfs quota {int(q)}/{int(p)}"
        except Exception:                               # noqa: BLE001
            pass
    return vis, eff, how


CPU_CAP = 16384                                           # oracle output capacity per pyramid (keeps allocation out of the timing)


def cpu_baseline(pyr_host, levels, budget_s=10.0, mt_s=3.0, log_bucket=0, bucket_limit=5, keep=0, kept=None):
    """The oracle (bit-exact plain-C restatement of the reference path, oracle/pislam_oracle.c) timed
    on ONE host thread — the reference itself is single-threaded — on a bounded sample of the same
    workload; then the same port on every visible host thread (pthreads inside liborc, one pyramid per
    thread at a time, timed in C) for >= mt_s seconds.  levels: (w, h, row0[, col0]).
    `kept` (a list): receives the oracle's (keypoints, descriptors) of the first `keep` pyramids — they are what
    parity_in_run compares the timed GPU outputs with, computed here anyway."""
    import numpy as np
    from oracle import orc
    orc.lib()
    okw = dict(log_bucket=log_bucket, bucket_limit=bucket_limit, cap=CPU_CAP)
    orc.pyramid4(pyr_host[0], levels, **okw)              # warm caches / lazy table
    n_kp, n_pyr, t0 = 0, 0, time.perf_counter()
    for b in range(64 * len(pyr_host)):                   # ~budget_s of work: the batch, repeated if need be
        kp, dsc, _ = orc.pyramid4(pyr_host[b % len(pyr_host)], levels, **okw)
        if kept is not None and b < keep and b < len(pyr_host):
            kept.append((kp, dsc))
        n_kp += len(kp)
        n_pyr += 1
        if time.perf_counter() - t0 > budget_s and (kept is None or len(kept) >= min(keep, len(pyr_host))):
            break
    dt = time.perf_counter() - t0
    vis, eff, how = effective_cores()
    out = {"value": n_kp / dt, "unit": "kp+desc/s", "cores": 1, "kind": "port",
           "sample": f"{n_pyr} pyramids of the batch ({n_kp} keypoints) in {dt:.2f} s, 1 thread, "
                     f"oracle/pislam_oracle.c -O3; host: {vis} CPUs visible, {eff:.1f} usable ({how})"}
    # reported beside the single-thread figure, which stays `value`
    try:
        nthr = max(1, vis)
        tot, done, dt2 = orc.pyramid_mt(np.ascontiguousarray(pyr_host), levels, nthr, mt_s, log_bucket=log_bucket,
                                        bucket_limit=bucket_limit, cap=CPU_CAP)
        out["all_threads"] = {"value": tot / dt2, "threads": nthr, "cores_visible": vis, "cores_effective": round(eff, 2),
                              "cores_effective_source": how,
                              "sample": f"{done} pyramids ({tot} keypoints) in {dt2:.2f} s, {nthr} pthreads (orc_pyramid_mt) "
                                        f"on {eff:.1f} usable CPUs"}
    except Exception as e:                                   # never let the extra leg break the bench line
        out["all_threads"] = {"error": repr(e)}
    return out


def parity_in_run(snap, kept, lane_hosts, levels, max_kp, log_bucket, bucket_limit, n_other=8, pins=None):
    """The proof inside the bench line: what the TIMED steps left in the lanes' output sets against the oracle
    (oracle/pislam_oracle.c) on the same pyramids — demo.cpp:85-113 prints its feature count; here the counts, the packed
    keypoints (order included) and the descriptor words are compared.
      snap[l] = (kp [n][max_kp], desc [n][max_kp][words], counts [B]) of lane l, copied to the host right after the timed
                region (numpy, uint32);
      kept    = the oracle's (keypoints, descriptors) of lane 0's first pyramids (computed by cpu_baseline's timing loop);
      lane_hosts[l] = the first pyramids of lane l on the host: lanes >= 1 are checked on `n_other` pyramids each (counts
                and keypoint words; lane 0 in full);
      pins    = optional reference counts every checked pyramid must have (the demo photo: 1754 / 1315)."""
    import numpy as np
    from oracle import orc
    full, kps, count_only, bad = 0, 0, 0, None

    def cmp(l, i, okp, odesc, want_desc):
        nonlocal kps, bad
        kp, desc, counts = snap[l]
        n = int(counts[i])
        if n != len(okp):
            bad = bad or f"lane {l} pyramid {i}: {n} keypoints, oracle {len(okp)}"
            return
        if pins is not None and n != pins:
            bad = bad or f"lane {l} pyramid {i}: {n} keypoints, the reference's pin is {pins}"
            return
        m = min(n, max_kp)
        if not np.array_equal(kp[i, :m], okp[:m]):
            bad = bad or f"lane {l} pyramid {i}: keypoint words differ from the oracle's"
            return
        if want_desc and not np.array_equal(desc[i, :m], odesc[:m]):
            bad = bad or f"lane {l} pyramid {i}: descriptor words differ from the oracle's"
            return
        kps += m

    for i, (okp, odesc) in enumerate(kept):
        if i < len(snap[0][0]):
            cmp(0, i, okp, odesc, True)
            full += 1
    for l in range(1, len(snap)):
        h = lane_hosts[l % len(lane_hosts)]
        memo = {}
        for i in range(min(n_other, len(snap[l][0]))):
            j = i % len(h)                                   # (a lane's batch tiles the pyramids it holds)
            if j not in memo:
                memo[j] = orc.pyramid4(h[j], levels, log_bucket=log_bucket, bucket_limit=bucket_limit, cap=CPU_CAP)
            cmp(l, i, memo[j][0], memo[j][1], True)
            count_only += 1
    return {"pyramids": full, "pyramids_other_lanes": count_only, "keypoints": kps, "ok": bad is None and full > 0,
            "compared": "keypoint count, packed keypoint words in order, descriptor words — oracle/pislam_oracle.c on the host "
                        "vs the output sets the timed pislam_pipeline_submit steps wrote (copied right after the timed region)",
            **({"reference_count_pin": pins} if pins is not None else {}),
            **({"first_mismatch": bad} if bad else {})}


# ---------------------------------------------------------------------------------------------------------
# other_workloads: every other BASELINE configuration, briefly, inside the driver's one line
# ---------------------------------------------------------------------------------------------------------
OTHER_WORKLOADS = [
    # name, bench.py arguments (besides the common ones)
    ("demo-photo", ["--workload", "demo-photo"]),                                  # natural content: the reference's own input
    ("1280x960", ["--workload", "1280x960", "--batch", "256", "--distinct", "16"]),  # configs[3] on spec (~2000 kp)
    ("1280x960-dense", ["--workload", "1280x960-dense", "--batch", "256", "--distinct", "16"]),   # what rounds 2-4 measured
    ("720p-build", ["--workload", "720p-build", "--batch", "64"]),                 # configs[4]
    ("vga-buckets43", ["--workload", "vga", "--log-bucket-size", "4", "--bucket-limit", "3", "--distinct", "32"]),   # README mode
]


def other_workloads(args, timeout_s: float = 90.0):
    """Each entry of OTHER_WORKLOADS as `python bench.py ...` (N=1, the same library and pipeline depth, --other-steps timed
    steps after a short clock ramp, parity_in_run on 8 pyramids of lane 0 + 4 of every other lane, no CPU leg): a compact
    record per workload.  Returns (records, all_parity_ok).  A child that fails or times out is recorded as an error —
    the headline line itself never depends on it — but a parity mismatch in a child fails the whole run (rc 6)."""
    import subprocess
    recs, ok = {}, True
    for name, extra in OTHER_WORKLOADS:
        cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(args.other_steps), "--warmup", "5", "--spin-s", "0.4",
               "--no-cpu-baseline", "--no-one-pyramid", "--no-other-workloads", "--parity-pyramids", "8",
               "--streams", str(args.streams), "--graph", str(args.graph)] + extra
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
            line = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
            if not line:
                recs[name] = {"error": f"rc {r.returncode}: {r.stderr.strip()[-300:]}"}
                continue
            d = json.loads(line[-1])
            pir = d.get("parity_in_run") or {}
            recs[name] = {"ms_per_step": d["ms_per_step"], "value": d["value"], "steps": d["steps"], "batch": d["config"]["batch_per_gpu"],
                          "keypoints_per_pyramid": d["config"]["keypoints_per_pyramid"],
                          "strips_us": d["roofline"]["launch_ms"] * 1e3, "strips_frac_of_hbm_peak": d["roofline"]["frac"],
                          "whole_step_frac": d["roofline"]["whole_step"]["frac"], "one_batch_ms": d.get("one_batch_ms"),
                          "parity_ok": bool(pir.get("ok")), "parity_pyramids": pir.get("pyramids"),
                          "parity_keypoints": pir.get("keypoints"), "baseline_config_index": d["config"]["baseline_config_index"],
                          "wall_s": round(time.perf_counter() - t0, 1), "rc": r.returncode}
            if r.returncode != 0 or not pir.get("ok"):
                ok = False
                recs[name]["first_mismatch"] = pir.get("first_mismatch")
        except subprocess.TimeoutExpired:
            recs[name] = {"error": f"timed out after {timeout_s:.0f} s"}
        except Exception as e:                               # noqa: BLE001
            recs[name] = {"error": repr(e)[:300]}
    return recs, ok


# ---------------------------------------------------------------------------------------------------------
# arguments
# ---------------------------------------------------------------------------------------------------------
def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--spin-s", type=float, default=0.5,
                    help="seconds of untimed load before the warm-up steps (GPU clock ramp)")
    ap.add_argument("--batch", type=int, default=256, help="pyramids per GPU")
    ap.add_argument("--distinct", type=int, default=0,
                    help="distinct synthetic pyramids generated per GPU (0 = all of the batch)")
    ap.add_argument("--shared-input", action="store_true",
                    help="all pipeline lanes read ONE input buffer (rounds 1-3; batch k+1's strips then stream the bytes batch k's "
                         "gather+ORB is still reading: cache hits a stream of distinct batches never gets).  Default: every lane "
                         "owns its batch — different pyramids (seeds), separate device buffers")
    ap.add_argument("--gen-workers", type=int, default=0,
                    help="processes generating the synthetic input (0 = the usable host cores, at most 16)")
    ap.add_argument("--max-keypoints", type=int, default=0,
                    help="keypoint / descriptor capacity per pyramid (0 = 4096 for vga, 8192 for the larger workloads)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="testing only (N=1): run the per-step count all-gather anyway, through a 1-rank RCCL communicator "
                         "created by the C ABI (pislam_dist_*), so that the N>1 data path is exercised on a 1-GPU box")
    ap.add_argument("--selftest-spawn", action="store_true",
                    help="testing only: exercise launch + supervision + rendezvous + count exchange with fake counts on "
                         "the CPU (no GPU work, value is null)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the 1-thread cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--one-pyramid", action="store_true", help=argparse.SUPPRESS)      # (rounds 1-4: opt-in; now the default)
    ap.add_argument("--no-one-pyramid", action="store_true",
                    help="skip one_pyramid_ms (ONE pyramid per call, the reference's frame-at-a-time use): profiled runs pass this "
                         "— its small launches would enter the kernel-trace averages and PMC medians of the batch kernels")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip other_workloads: by default the N=1 run of the headline workload ends with a SHORT run of every other "
                         "configuration (the reference's demo photo, configs[3] on spec and at the dense shape count, configs[4], the "
                         "README bucket mode <4,3>) as child processes of this one — ms/step, strip-kernel time, keypoints per pyramid and "
                         "parity_in_run of each — so that the driver's ONE line carries a number for every BASELINE config")
    ap.add_argument("--other-steps", type=int, default=30, help="timed steps of each other_workloads child")
    ap.add_argument("--dist-backend", default=None,
                    help="override the torch.distributed backend (default nccl = RCCL); 'gloo' lets the N>1 code "
                         "path be exercised with several ranks sharing one GPU (testing only)")
    ap.add_argument("--exchange", default="cabi", choices=["cabi", "torch"],
                    help="N>1 count all-gather: the C ABI's communicator (pislam_dist_*, default) or torch.distributed's")
    ap.add_argument("--workload", default="vga", choices=["vga", "1280x960", "1280x960-dense", "720p-build", "demo-photo"],
                    help="vga = BASELINE configs[1] (default, the headline; configs[2] when N>1); 1280x960 = configs[3] "
                         "(packed layout, vstep 1280, ~2000 keypoints per pyramid as BASELINE.json names it); 1280x960-dense = "
                         "the same layout with the VGA shape density (~4400 keypoints per pyramid: rounds 2-4 measured this); "
                         "720p-build = configs[4] (gaussian5x5 + bilinear pyramid build on the GPU inside the timed step); "
                         "demo-photo = the reference's own demo pyramid (tests/golden/demo_pyramid.npz: natural image content, "
                         "1754 keypoints) x batch, every lane its own copy")
    ap.add_argument("--parity-pyramids", type=int, default=32,
                    help="parity_in_run: pyramids of lane 0 whose timed outputs (counts, keypoints, descriptors) are compared with "
                         "the oracle after the timed region (other lanes: 8 each); 0 = off.  A mismatch ends the run with rc 6")
    ap.add_argument("--pipeline", type=int, default=0, help="0 auto (fused), 1 staged, 2 fused")
    ap.add_argument("--strip-rows", type=int, default=0)
    ap.add_argument("--orb-chunks", type=int, default=0)
    ap.add_argument("--wgs-per-cu", type=int, default=0)
    ap.add_argument("--run-len", type=int, default=0)
    ap.add_argument("--alias", type=int, default=-1)
    ap.add_argument("--run-order", type=int, default=-1, help="1 (default): a pyramid's runs are launched longest first; 0: entry order")
    ap.add_argument("--strip-px", type=int, default=0, help="profiling: pixels per strip the height heuristic aims at (default 16384)")
    ap.add_argument("--strip-rows-max", type=int, default=0, help="profiling: upper bound of the heuristic strip height (default: 56 for large launches, 28 for small ones)")
    ap.add_argument("--tile-cols", type=int, default=0, help="levels with more classified columns are cut into x-tiles (0 = default 704, < 0 never)")
    ap.add_argument("--orb-in-strip", type=int, default=-1, help="1: strips describe their own keypoints; 0 (default): one gather+ORB pass")
    ap.add_argument("--graph", type=int, default=1,
                    help="1 (default): the library pipeline replays a repeated batch call from a hipGraph (pislam_pipeline option "
                         "'graphs') — falls back to eager launches if a capture or the replay check fails; 0: eager")
    ap.add_argument("--streams", type=int, default=0,
                    help="lanes of the library pipeline (pislam_pipeline_create depth: HIP stream + context each); step k runs on lane k %% S, so "
                         "consecutive batches overlap on the GPU.  0 = default (3); 1 = strictly one batch call at a time")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="any library option (pislam_ctx_set_option), e.g. --opt sub_batches=2; repeatable")
    ap.add_argument("--margins-clean", type=int, default=1,
                    help="720p-build: 1 (default) the per-step rebuild vouches for the zero margins its own first fill left "
                         "(PISLAM_BUILD_MARGINS_CLEAN: the steady state of a stream refilling one buffer); 0 re-zeroes them every step")
    ap.add_argument("--match", action="store_true",
                    help="also match every pyramid's descriptors against its neighbour's inside the step (SURVEY 8f-4)")
    ap.add_argument("--match-mfma", type=int, default=-1, help="--match: 1 (default) matrix-core matcher, 0 the VALU popcount kernel")
    ap.add_argument("--lds-pad", type=int, default=0, help="profiling only: extra LDS per strip workgroup")
    ap.add_argument("--log-bucket-size", type=int, default=0, help="fastExtract logBucketSize (README uses 4)")
    ap.add_argument("--bucket-limit", type=int, default=5, help="fastExtract bucketLimit (README uses 3)")
    ap.add_argument("--ablate", type=int, default=0, help="profiling only (results invalid): fused-kernel phase mask")
    # ---- supervision of the N>1 run ----
    ap.add_argument("--wall-cap", type=float, default=900.0,
                    help="wall-clock cap in seconds for the whole run incl. every fallback attempt (rc != 0 beyond it)")
    ap.add_argument("--attempt-timeout", type=float, default=120.0,
                    help="N>1: hard timeout of one worker attempt (a healthy attempt takes ~40 s; two retries fit the wall cap)")
    ap.add_argument("--selfcheck-timeout", type=float, default=60.0,
                    help="N>1 worker: hard timeout of the start-up self-check (2 untimed steps per pipeline incl. the exchange)")
    ap.add_argument("--worker", action="store_true", help=argparse.SUPPRESS)            # a supervisor's child
    ap.add_argument("--attempt", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--dropped", default="[]", help=argparse.SUPPRESS)                  # JSON list: what earlier attempts dropped
    return ap


# ---------------------------------------------------------------------------------------------------------
# N > 1: supervisors
# ---------------------------------------------------------------------------------------------------------
# The ladder: (what this rung DROPS relative to the previous one, argument overrides — cumulative)
LADDER = [
    (None, {}),
    ("hipGraph replay -> eager launches", {"graph": 0}),
    ("batches in flight on several streams -> one stream, one batch call at a time", {"streams": 1}),
    ("C-ABI RCCL communicator (pislam_dist_*) -> torch.distributed all-gather", {"exchange": "torch"}),
    ("RCCL -> counts all-gathered on the host through gloo (no GPU collective)", {"dist_backend": "gloo"}),
]


def _inject(stage: str, attempt: int, rank: int):
    """Test hook: PISLAM_BENCH_INJECT="<attempt>:<stage>:<mode>[:<rank>],..." with stage in {start, selfcheck, run} and
    mode in {hang, fail, exit}: lets the CPU tests break a chosen attempt at a chosen stage and watch the ladder."""
    spec = os.environ.get("PISLAM_BENCH_INJECT", "")
    for item in filter(None, spec.split(",")):
        f = item.split(":")
        if int(f[0]) != attempt or f[1] != stage or (len(f) > 3 and int(f[3]) != rank):
            continue
        if f[2] == "hang":
            time.sleep(10 ** 6)
        if f[2] == "exit":
            os._exit(7)
        raise RuntimeError(f"injected failure at attempt {attempt}, stage {stage}")


def _spin_seconds(spin_s: float, rank: int) -> float:
    """Test hook: PISLAM_BENCH_SPIN_SKEW="<rank>:<factor>[,...]" multiplies --spin-s on chosen ranks, so that the CPU
    tests can make the ranks' own clocks disagree about the length of the ramp."""
    for item in filter(None, os.environ.get("PISLAM_BENCH_SPIN_SKEW", "").split(",")):
        r, _, f = item.partition(":")
        if int(r) == rank:
            spin_s *= float(f)
    return spin_s


def clock_ramp(spin_s: float, group, agree_any) -> int:
    """Untimed load for about `spin_s` seconds with the SAME number of groups on every rank.

    `group()` may issue collectives (every step of an N>1 run ends with the count all-gather), so the decision to run
    another group cannot be taken from a rank's own clock: before each group every rank says whether ITS clock wants
    more, the flags are all-reduced (MAX, control plane), and all ranks run the group or all leave — the ramp lasts
    until the slowest rank's spin_s has passed.  Returns the number of groups run (equal on every rank)."""
    t0, groups = time.perf_counter(), 0
    while agree_any(time.perf_counter() - t0 < spin_s):
        group()
        groups += 1
    return groups


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def supervisor_main(args, argv):
    """One per torchrun rank.  CPU only: starts the rank's worker, enforces timeouts, walks the ladder."""
    import datetime
    import signal
    import subprocess
    import tempfile
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    deadline = time.monotonic() + args.wall_cap
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    child = {"p": None}

    def kill_child():
        p = child["p"]
        if p is not None and p.poll() is None:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except Exception:                           # noqa: BLE001
                pass
            try:
                p.wait(timeout=10)
            except Exception:                           # noqa: BLE001
                pass

    def on_term(signum, frame):                         # torchrun tearing the job down: take the worker with us
        kill_child()
        os._exit(128 + signum)

    signal.signal(signal.SIGTERM, on_term)
    signal.signal(signal.SIGINT, on_term)

    def agree_max(v: int) -> int:
        t = torch.tensor([v], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return int(t.item())

    cfg = {"graph": args.graph, "streams": args.streams, "exchange": args.exchange, "dist_backend": args.dist_backend}
    dropped, tried, attempt, history = [], set(), 0, []
    rc_final, line = 3, None
    tmpdir = tempfile.mkdtemp(prefix="pislam_bench_")
    for label, over in LADDER:
        new = dict(cfg, **over)
        if label is not None and new == cfg:            # this rung drops nothing the configuration still has
            continue
        cfg = new
        if label is not None:
            dropped.append(label)
        key = json.dumps(cfg, sort_keys=True)
        if key in tried:
            continue
        tried.add(key)
        if agree_max(1 if deadline - time.monotonic() < 30 else 0):
            history.append("wall-clock cap reached before the next attempt")
            break
        # a rendezvous port of its own for this attempt's workers (rank 0 draws it)
        port = torch.tensor([_free_port() if rank == 0 else 0], dtype=torch.int64)
        dist.broadcast(port, src=0)
        wargs = list(argv) + ["--worker", "--attempt", str(attempt), "--dropped", json.dumps(dropped), "--graph", str(cfg["graph"]),
                              "--streams", str(cfg["streams"]), "--exchange", cfg["exchange"]]
        if cfg["dist_backend"]:
            wargs += ["--dist-backend", cfg["dist_backend"]]
        # (the workers rendezvous among themselves on the fresh port: rank 0 hosts the store — torchrun's
        #  TORCHELASTIC_USE_AGENT_STORE would make it look for the agent's store there instead)
        env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
        env.update(MASTER_PORT=str(int(port.item())), MASTER_ADDR="127.0.0.1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out_path = os.path.join(tmpdir, f"attempt{attempt}_rank{rank}.out")
        timeout = max(10.0, min(args.attempt_timeout, deadline - time.monotonic() - 15))
        with open(out_path, "w") as fout:
            p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + wargs, env=env, stdout=fout,
                                 stderr=None, start_new_session=True, cwd=os.getcwd())
            child["p"] = p
            t0 = time.monotonic()
            while True:
                rc = p.poll()
                if rc is not None:
                    status = "ok" if rc == 0 else f"worker exit code {rc}"
                    break
                if time.monotonic() - t0 > timeout:
                    status = f"worker still running after {timeout:.0f} s: killed"
                    kill_child()
                    break
                time.sleep(0.1)
        bad = agree_max(0 if status == "ok" else 1)
        if not bad:
            if rank == 0:
                lines = [ln for ln in open(out_path).read().splitlines() if ln.startswith("{")]
                line = lines[-1] if lines else None
            if not agree_max(0 if (rank != 0 or line) else 1):
                rc_final = 0
                break
            status = "rank 0's worker printed no JSON line"
        kill_child()                                    # every rank's worker of a failed attempt goes away
        history.append(f"attempt {attempt} ({'as configured' if not dropped else 'after dropping: ' + dropped[-1]}): "
                       f"rank {rank}: {status}")
        if rank == 0:
            print(f"[bench supervisor] {history[-1]}{'; a rank failed' if bad else ''} -> next rung", file=sys.stderr, flush=True)
        attempt += 1
    if rank == 0:
        if rc_final == 0:
            print(line, flush=True)
        else:
            print(f"[bench supervisor] no configuration completed (wall cap {args.wall_cap:.0f} s): " + " | ".join(history),
                  file=sys.stderr, flush=True)
    try:
        dist.barrier()
        dist.destroy_process_group()
    except Exception:                                   # noqa: BLE001
        pass
    return rc_final


# ---------------------------------------------------------------------------------------------------------
# the worker (the whole bench for N = 1)
# ---------------------------------------------------------------------------------------------------------
class Watchdog:
    """Hard timeout for a phase that can hang inside a C call (a wedged collective, a stuck stream sync): Python
    signal handlers do not run there, a daemon thread does.  On expiry: message, stack dump, os._exit(rc)."""

    def __init__(self, seconds: float, what: str, rc: int = 4):
        self.t = threading.Timer(seconds, self._fire)
        self.t.daemon = True
        self.what, self.rc, self.seconds = what, rc, seconds

    def _fire(self):
        import faulthandler
        print(f"[bench worker rank {os.environ.get('RANK', '0')}] {self.what} did not finish within {self.seconds:.0f} s: aborting",
              file=sys.stderr, flush=True)
        try:
            faulthandler.dump_traceback(file=sys.stderr)
        except Exception:                               # noqa: BLE001
            pass
        os._exit(self.rc)

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *exc):
        self.t.cancel()
        return False


def load_counters(workload: str, batch: int, suffix: str = ""):
    """PMC passes cannot run inside this process: HBM bytes and instruction counts per step come from the
    committed profile of this workload (tools/profile_round.sh -> profiles/<tag>_counters_<workload><suffix>.json;
    suffix "_buckets43" = the README bucket mode, "_streams3" = the pass taken with three batches in flight) and are
    used ONLY when that profile was taken on exactly these kernel sources and this shape."""
    from pislam_amd import build as pbuild
    path = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_counters_{workload}{suffix}.json")
    rel = os.path.relpath(path, ROOT)
    if not os.path.exists(path):
        return None, f"null: no PMC profile for this workload / configuration ({rel})"
    try:
        tj = json.load(open(path))
    except Exception as e:                              # noqa: BLE001
        return None, f"null: {e!r}"
    if tj.get("source_hash") != pbuild.source_hash():
        return None, f"null: {rel} was measured on kernel sources {tj.get('source_hash')}, this run uses {pbuild.source_hash()}"
    if tj.get("batch") != batch:
        return None, f"null: {rel} was measured at batch {tj.get('batch')}"
    return tj, f"{rel} (rocprofv3 --pmc passes on kernel sources {tj['source_hash']}: {tj.get('command', '')}; not measured in this run)"


def worker_main(args):
    import numpy as np
    import torch
    from pislam_amd import dist as pdist

    selftest = args.selftest_spawn
    dropped = json.loads(args.dropped)
    _inject("start", args.attempt, int(os.environ.get("RANK", "0")))
    rank, local_rank, world = pdist.init(backend="gloo" if selftest else args.dist_backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                         "(or run `python bench.py --gpus N` without a torchrun environment)")

    def agree_any(flag: bool, dev=None) -> bool:
        """True on every rank if `flag` is true on any rank (control plane)."""
        if world == 1:
            return flag
        import torch.distributed as dist
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev if dist.get_backend() == "nccl" else None)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(int(t.item()))

    if selftest:
        # launch / supervision / rendezvous / exchange plumbing only (CPU, gloo): every rank contributes fake counts
        B = 4
        fake = torch.arange(B, dtype=torch.int32) + 100 * rank
        xchg = pdist.CountExchange(world)
        with Watchdog(args.selfcheck_timeout, "start-up self-check"):
            _inject("selfcheck", args.attempt, rank)
            for _ in range(2):
                xchg.before_step()
                xchg.start(fake)
            xchg.finish()
        _inject("run", args.attempt, rank)

        def fake_group():                                 # what spin_once() x 8 + synchronize is to the real worker
            for _ in range(8):
                xchg.before_step()
                xchg.start(fake)
            xchg.finish()
            time.sleep(0.002)

        ramp_groups = clock_ramp(_spin_seconds(args.spin_s, rank), fake_group, agree_any)
        xchg.before_step()
        xchg.start(fake)
        allc = xchg.finish()
        ok = allc.tolist() == [100 * r + i for r in range(world) for i in range(B)]
        coll_ok, coll = pdist.collectives_agree([xchg], world)
        if world > 1:
            torch.distributed.barrier()
        if not coll_ok:
            raise SystemExit(f"ranks issued different numbers of count all-gathers: {coll}")
        if rank == 0:
            print(json.dumps({"metric": METRIC, "value": None, "unit": "kp+desc/s", "n_gpus": world, "selftest": "spawn",
                              "exchange_ok": ok, "count_allgather": xchg.path,
                              "config": {"dist_fallbacks": dropped, "rccl_ranks": None, "attempt": args.attempt,
                                         "graph": args.graph, "streams": args.streams, "exchange": args.exchange,
                                         "dist_backend": args.dist_backend, "ramp_groups": ramp_groups,
                                         "count_allgathers_per_rank": coll}}), flush=True)
        if world > 1:
            torch.distributed.destroy_process_group()
        return 0

    from pislam_amd import synth
    from pislam_amd.frontend import OrbFrontend
    from pislam_amd.capi import Context

    gloo_mode = args.dist_backend == "gloo"
    if gloo_mode:
        local_rank = local_rank % max(1, torch.cuda.device_count())   # ranks may share a GPU in this mode
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    nshapes = None                                          # synth.make_level0's rule: 80 shapes per VGA frame area
    if args.workload in ("vga", "demo-photo"):
        levels = synth.level_table()                       # demo.cpp:38-47 table, 2210 stacked rows
        vstep, w0, h0 = 640, 640, 480
    elif args.workload in ("1280x960", "1280x960-dense"):
        levels = synth.packed_level_table(1280, 960)       # 3768 rows: levels 4|5 and 6|7 side by side
        vstep, w0, h0 = 1280, 1280, 960
        # BASELINE.json configs[3] / SURVEY 8d name "~2000 kp/frame": 148 shapes per frame give 1990-2060 keypoints per
        # pyramid at threshold 20 (calibrated with the oracle); the area-scaled density (320 shapes, ~4400 keypoints) that
        # rounds 2-4 measured stays available as 1280x960-dense
        nshapes = 148 if args.workload == "1280x960" else None
    else:
        from pislam_amd.frontend import PyramidBuilder
        w0, h0 = 1280, 720
    B = args.batch
    distinct = args.distinct or (min(B, 16) if args.workload == "720p-build" else 1 if args.workload == "demo-photo" else B)
    first = rank * B
    S = 1 if args.match else (args.streams if args.streams > 0 else 3)
    force = args.force_exchange and world == 1
    # Output sets per pipeline: two whenever counts are exchanged, so that a pipeline's next step (into the other
    # set) does not wait for the all-gather of its previous one — the collective's latency (event hand-over to the
    # collective stream + a latency-bound RCCL kernel, ~50 us) would otherwise be added to every pipeline cycle.
    nsets = 2 if (world > 1 or force) else 1
    fallbacks = list(dropped)                              # what the supervisors' ladder dropped + what this worker drops
    extra_opts = []
    for kv in args.opt:
        k, _, v = kv.partition("=")
        extra_opts.append((k.strip(), int(v)))

    lib_opts = [("pipeline", args.pipeline), ("strip_rows", args.strip_rows), ("orb_chunks", args.orb_chunks),
                ("lds_pad", args.lds_pad)]
    for key, val, on in (("alias", args.alias, args.alias >= 0), ("run_order", args.run_order, args.run_order >= 0),
                         ("strip_px", args.strip_px, args.strip_px), ("strip_rows_max", args.strip_rows_max, args.strip_rows_max),
                         ("tile_cols", args.tile_cols, args.tile_cols), ("orb_in_strip", args.orb_in_strip, args.orb_in_strip >= 0),
                         ("match_mfma", args.match_mfma, args.match_mfma >= 0), ("run_len", args.run_len, args.run_len),
                         ("wgs_per_cu", args.wgs_per_cu, args.wgs_per_cu)):
        if on:
            lib_opts.append((key, val))
    lib_opts += extra_opts + [("ablate", args.ablate)]

    # ---- the resident input: every lane owns ITS batch — lane l of rank r holds the pyramids (seeds) number
    # (r * S + l) * B ... — in a device buffer of its own (--shared-input: rounds 1-3, one buffer read by all lanes).
    # The 720p workload builds its pyramids per lane from the lane's own frames.
    nd = min(distinct, B)
    nlane_in = 1 if args.shared_input else S
    gen_workers = args.gen_workers or max(1, min(16, int(effective_cores()[1]) // max(1, min(world, 8))))
    # (the generator takes ~60 ms per VGA pyramid and host thread: on a host with few usable cores per rank the input is
    #  built from fewer different pyramids, each lane's batch tiling them — the bench must start within its attempt timeout)
    if not args.distinct:
        px = w0 * h0 / (640.0 * 480.0)
        while nd > 16 and nd * nlane_in * 0.06 * px / gen_workers > 40.0:
            nd //= 2
    idx = [(rank * S + l) * B + i for l in range(nlane_in) for i in range(nd)]
    host = d_frames = None
    lane_hosts = []                                       # per input lane: its first pyramids on the host (parity_in_run)
    lane_in = []                                          # per input lane: device tensor (frames or pyramids)
    if args.workload == "demo-photo":
        # the reference's own demo input (data fixture; the image its README and demo.cpp run on), `B` copies per lane in
        # a device buffer of the lane's own: natural image content, 1754 keypoints per pyramid (1315 in README mode <4,3>)
        rows = synth.pyramid_rows(levels)
        img = np.load(os.path.join(ROOT, "tests", "golden", "demo_pyramid.npz"))["img"]
        assert img.shape == (rows, vstep)
        host = np.ascontiguousarray(img[None])
        nd = 1
        for l in range(nlane_in):
            lane_in.append(torch.from_numpy(host).to(dev).expand(B, rows, vstep).contiguous())
            lane_hosts.append(host)
    elif args.workload == "720p-build":
        fr = synth.make_many(idx, workers=gen_workers, kind="level0", w0=w0, h0=h0)
        for l in range(nlane_in):
            t = torch.from_numpy(fr[l * nd:(l + 1) * nd]).to(dev)
            lane_in.append(t[torch.arange(B, device=dev) % nd].contiguous() if nd < B else t)
        d_frames = lane_in[0]
    else:
        rows = synth.pyramid_rows(levels)
        hp = synth.make_many(idx, workers=gen_workers, w0=w0, h0=h0, vstep=vstep, levels=levels, nshapes=nshapes)
        host = hp[:nd]                                    # lane 0's batch: the cpu_baseline sample
        for l in range(nlane_in):
            t = torch.from_numpy(hp[l * nd:(l + 1) * nd]).to(dev)
            lane_in.append(t[torch.arange(B, device=dev) % nd].contiguous() if nd < B else t)
            lane_hosts.append(hp[l * nd:l * nd + min(8, nd)].copy())
        del hp

    # ---- batches in flight: the LIBRARY's pipeline object (pislam_pipeline_*, include/pislam_hip.h) — S lanes, each a
    # context (workspace) + non-blocking HIP stream of its own; batch k is submitted to lane k % S, so that the tail of
    # one batch call runs under the head of the next; a call that repeats exactly is replayed from a hipGraph inside
    # the library (option "graphs").  Every batch is still processed completely (strips -> overflow pass ->
    # gather+ORB, then the count all-gather) inside the timed region.
    from pislam_amd import capi
    pl = capi.Pipeline(device=local_rank, depth=S)
    pl.set_option("graphs", 1 if args.graph else 0)
    for k_, v_ in lib_opts:
        pl.set_option(k_, v_)

    class Lane:
        pass

    pipes = []
    for i in range(S):
        P = Lane()
        P.ctx = pl.lane(i)                                # borrowed: the pipeline owns it
        P.stream_handle = pl.stream_of(i)                 # (ticket i runs on lane i)
        P.stream = torch.cuda.ExternalStream(P.stream_handle, device=dev)
        P.builder = None
        P.src = lane_in[i % nlane_in]                     # this lane's frames (720p-build) or pyramids
        with torch.cuda.stream(P.stream):
            if args.workload == "720p-build":
                P.builder = PyramidBuilder(w0, h0, ctx=P.ctx)
                levels, vstep, rows = P.builder.levels, P.builder.vstep, P.builder.rows
                P.d_pyr = torch.empty((B, rows, vstep), dtype=torch.uint8, device=dev)
                P.builder(P.src, P.d_pyr)
                torch.cuda.synchronize()
                if host is None:
                    host = P.d_pyr[:nd].cpu().numpy()
                lane_hosts.append(P.d_pyr[:8].cpu().numpy())   # (the pyramids the GPU built: the front-end's own input)
            else:
                P.d_pyr = P.src
        pipes.append(P)
    fe = OrbFrontend(levels, vstep=vstep, rows=rows, max_keypoints=args.max_keypoints, ctx=pipes[0].ctx,
                     log_bucket_size=args.log_bucket_size, bucket_limit=args.bucket_limit)   # parameter / level structs
    pl.reserve(fe.params, fe.levels, B)
    for P in pipes:
        P.outs = [fe.alloc_outputs(B, dev) for _ in range(nsets)]
    # one plain context for everything measured OUTSIDE the timed region (one call at a time, the strip kernel alone)
    side_stream = torch.cuda.Stream(dev)
    ctx = Context(device=local_rank, stream=side_stream.cuda_stream)
    for k_, v_ in lib_opts:
        ctx.set_option(k_, v_)
    fe1 = OrbFrontend(levels, vstep=vstep, rows=rows, max_keypoints=args.max_keypoints, ctx=ctx,
                      log_bucket_size=args.log_bucket_size, bucket_limit=args.bucket_limit)
    fe1.reserve(B)
    builder1 = PyramidBuilder(w0, h0, ctx=ctx) if args.workload == "720p-build" else None
    builder = pipes[0].builder
    d_pyr, stream = pipes[0].d_pyr, side_stream
    kp, desc, counts = fe.alloc_outputs(B, dev)

    # ---- the count all-gather: ONE communicator and ONE collective stream per process, shared by all lanes
    # (pislam_amd.dist.ExchangeHub over the C ABI's pislam_dist_*; the all-gathers are ordered after / fence the lanes'
    # streams).  Ranks sharing one GPU (gloo test mode) cannot form an RCCL communicator; if the C-ABI path fails on
    # any rank, all ranks fall back to torch.distributed's all-gather and the JSON line says so.
    hub = None
    if world > 1 and not gloo_mode and args.exchange == "cabi":
        err = pdist.init_rccl(ctx, rank, world, dev)
        if err:
            fallbacks.append(f"C-ABI RCCL communicator unavailable ({err}) -> torch.distributed all-gather")
            if rank == 0:
                print(f"[bench] C-ABI RCCL path unavailable, using torch.distributed: {err}", file=sys.stderr)
        else:
            hub = pdist.ExchangeHub(ctx)
    if force:
        ctx.set_option("dist_rccl_single", 1)
        ctx.dist_init(capi.dist_unique_id(), 0, 1)
        hub = pdist.ExchangeHub(ctx)
    for P in pipes:
        P.xchg = pdist.CountExchange(world, hub=hub, always_collective=force, sets=nsets, stream=P.stream_handle)
    xchg = pipes[0].xchg
    rccl_ranks = None
    if hub is not None:
        rccl_ranks = ctx.dist_comm_count()                 # what RCCL itself says (ncclCommCount)
    elif world > 1 and not gloo_mode:
        rccl_ranks = torch.distributed.get_world_size()    # torch's RCCL process group

    m_out = None
    if args.match:
        # train side: the neighbouring pyramid's descriptors (static inputs -> made once, outside the step)
        from pislam_amd.frontend import matchHammingBatch
        with torch.cuda.stream(stream):
            fe1(d_pyr, kp, desc, counts)
        torch.cuda.synchronize()
        t_desc, t_counts = torch.roll(desc, 1, 0).contiguous(), torch.roll(counts, 1, 0).contiguous()
        m_out = [torch.empty((B, args.max_keypoints), dtype=torch.int32, device=dev) for _ in range(3)]
    nstep, tick, last_out = [0], [0], [None]

    def step():
        """Batch k: on lane k % S — [pyramid build on the lane's stream] -> pislam_pipeline_submit -> [matcher] ->
        count all-gather ordered after the lane's stream."""
        k = tick[0]                                     # == the library's ticket: lane k % S
        tick[0] += 1
        nstep[0] += 1
        P = pipes[k % S]
        o = P.outs[(k // S) % nsets]
        last_out[0] = o
        with torch.cuda.stream(P.stream):
            P.xchg.before_step()                        # the lane's stream waits for the all-gather that read this output set
            if P.builder is not None:
                # steady state of a stream: the same pyramid buffer is refilled every step by the same builder
                # (its first fill, before the timed region, established the zero margins)
                P.builder(P.src, P.d_pyr, margins_clean=bool(args.margins_clean))
            t = pl.submit(fe.params, fe.levels, P.d_pyr, *o)
            assert t == k, "tickets and steps out of phase"
            if m_out is not None:
                matchHammingBatch(o[1], o[2], t_desc, t_counts, *m_out, ctx=P.ctx)
            P.xchg.start(o[2])
        return P

    def launches1(k_, d_, c_):
        """One call at a time on the plain context (outside the timed region)."""
        if builder1 is not None:
            builder1(d_frames, d_pyr, margins_clean=bool(args.margins_clean))
        fe1(d_pyr, k_, d_, c_)
        if m_out is not None:
            matchHammingBatch(d_, c_, t_desc, t_counts, *m_out, ctx=ctx)

    def finish_all(last=None):
        r = None
        for P in pipes:                                 # every step's all-gather has completed
            v = P.xchg.finish()
            if P is last:
                r = v
        return r

    # ---- warm-up of the lanes: every (lane, output set) call three times — eager, captured, first replay — and the
    # replayed result checked against a plain eager call: the measurement must never depend on the graphs
    for _ in range(3 * S * nsets):
        step()
    finish_all()
    torch.cuda.synchronize()
    graph_bad = False
    for P in reversed(pipes):                             # (lane 0 last: `counts` then holds lane 0's, as launches1 would leave them)
        with torch.cuda.stream(stream):
            fe1(P.d_pyr, kp, desc, counts)                # plain eager call on the lane's own (already built) pyramids
        torch.cuda.synchronize()
        graph_bad = graph_bad or any(not torch.equal(o[2], counts) for o in P.outs)
    st0 = pl.stats()
    use_graphs = bool(args.graph) and st0["replayed_from_graphs"] > 0
    if agree_any(graph_bad, dev):
        # every rank takes the same path (a rank replaying graphs beside ranks launching eagerly would be a
        # different measurement per rank)
        pl.set_option("graphs", 0)
        use_graphs = False
        fallbacks.append("a hipGraph replay did not reproduce the eager counts on a rank -> eager launches")
    elif args.graph and agree_any(st0["capture_failed"] > 0 or st0["replayed_from_graphs"] == 0, dev):
        pl.set_option("graphs", 0)
        use_graphs = False
        fallbacks.append("hipGraph capture failed on a rank -> eager launches")
    nstep[0] = 0

    def spin_once():
        for _ in range(S):
            step()

    # ---- start-up self-check (N > 1): two untimed steps per pipeline INCLUDING the exchange, under a hard timeout
    # — a collective that cannot complete must end this worker (the supervisors then walk the ladder), not the
    # driver's one shot at the 8-GPU run.
    if world > 1 or force:
        with Watchdog(args.selfcheck_timeout, "start-up self-check (2 steps per pipeline + count all-gather)"):
            _inject("selfcheck", args.attempt, rank)
            Pl = None
            for _ in range(2 * S):
                Pl = step()
            chk = finish_all(Pl)
            torch.cuda.synchronize()
            mine = last_out[0][2]
            bad = not torch.equal(chk[rank * B:(rank + 1) * B].cpu(), mine.cpu())
            if agree_any(bad, dev):
                raise SystemExit("start-up self-check: the gathered counts do not hold this rank's counts")
        nstep[0] = 0

    # Clock ramp: the GPU idles at a few hundred MHz and needs a fraction of a second of load to reach its
    # sustained clocks — far longer than a handful of 0.4 ms steps.  Spin the same step, untimed, before the W
    # warm-up steps so that W and K measure the steady state whatever their values.  Every step of an N>1 run issues
    # a count all-gather, so the number of ramp groups is AGREED between the ranks (clock_ramp), never taken from a
    # rank's own clock: ranks that disagreed by one group would differ by 8*S collectives and deadlock.
    _inject("run", args.attempt, rank)

    def ramp_group():
        for _ in range(8):
            spin_once()
        torch.cuda.synchronize()

    ramp_groups = clock_ramp(_spin_seconds(args.spin_s, rank), ramp_group, lambda f: agree_any(f, dev))
    for _ in range(args.warmup):
        step()
    finish_all()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    nstep[0] = 0
    tick_timed = [tick[0]]                                 # ticket of the first timed step
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
    allc = finish_all(last)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # parity_in_run: what the timed steps left in the lanes' output sets (the set the lane's LAST timed step wrote), copied
    # now — before anything else touches the pipelines
    snap = None
    NP = max(0, args.parity_pyramids)
    if NP:
        snap = []
        for li, P in enumerate(pipes):
            last_k = max((k for k in range(tick_timed[0], tick_timed[0] + args.steps) if k % S == li), default=None)
            o = P.outs[(last_k // S) % nsets] if last_k is not None else P.outs[0]
            n = min(B, NP if li == 0 else 8)
            snap.append((o[0][:n].cpu().numpy().view(np.uint32), o[1][:n].cpu().numpy().view(np.uint32).reshape(n, args.max_keypoints, -1),
                         o[2].cpu().numpy().view(np.uint32)))
    clock_ghz = None
    try:
        for _ in range(4):
            spin_once()                                  # (the probe wave shares the GPU with the steps' kernels)
        side = Context(device=local_rank)
        side.set_option("own_stream", 2)
        clock_ghz = side.shader_clock_ghz(200)
        side.close()
        torch.cuda.synchronize()
    except Exception:                                    # noqa: BLE001
        pass
    dt_rank = dt
    dts = [dt]
    coll_ok, coll = pdist.collectives_agree([P.xchg for P in pipes], world, dev)   # outside the timed region
    if not coll_ok:
        raise SystemExit(f"ranks issued different numbers of count all-gathers: {coll}")
    if world > 1:
        import torch.distributed as dist
        on_gpu = dist.get_backend() == "nccl"
        t = torch.tensor([dt], dtype=torch.float64, device=dev if on_gpu else None)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        dts = [float(x.item()) for x in allt]
        dt = max(dts)                                    # the contract: MAX over ranks

    def local_kp_fn(c_):
        return int(torch.clamp(c_.to(torch.int64), max=args.max_keypoints).sum().item())

    # ---- one batch call at a time (the figure a caller without any stream choreography gets): a pipeline of depth 1
    # (same library path, same graphs), and the stage times of eager calls on a plain context ----
    ev = []
    for _ in range(min(8, max(3, args.steps))):
        with torch.cuda.stream(stream):
            launches1(kp, desc, counts)
        ev.append(fe1.last_timing())
    ev_total_ms = float(np.mean([e[0] for e in ev]))
    ev_stage_ms = [float(np.mean([e[1][i] for e in ev])) for i in range(3)]
    one_ms = None
    one_pyr_ms = one_pyr_lat_ms = one_pyr_path = None
    try:
        pl1 = capi.Pipeline(device=local_rank, depth=1)
        pl1.set_option("graphs", 1 if use_graphs else 0)
        for k_, v_ in lib_opts:
            pl1.set_option(k_, v_)
        pl1.reserve(fe.params, fe.levels, B)
        c1 = pl1.lane(0)
        s1 = torch.cuda.ExternalStream(pl1.stream_of(0), device=dev)
        b1 = PyramidBuilder(w0, h0, ctx=c1) if builder is not None else None
        o1 = fe.alloc_outputs(B, dev)
        REPS = 30

        def one_call():
            if b1 is not None:
                b1(d_frames, d_pyr, margins_clean=bool(args.margins_clean))
            pl1.submit(fe.params, fe.levels, d_pyr, *o1)
            if m_out is not None:
                matchHammingBatch(o1[1], o1[2], t_desc, t_counts, *m_out, ctx=c1)

        brackets = []
        with torch.cuda.stream(s1):
            for _ in range(4):                           # eager, captured, first replays
                one_call()
            for _ in range(3):                           # three brackets, the fastest counts (clock ramps, stragglers)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s1)
                for _ in range(REPS):
                    one_call()
                e1.record(s1)
                brackets.append((e0, e1))
        torch.cuda.synchronize()
        one_ms = min(a.elapsed_time(b) for a, b in brackets) / REPS
    except Exception as e:                               # noqa: BLE001
        pl1 = None
        print(f"[bench] one-call-at-a-time measurement failed: {e!r}", file=sys.stderr)
    lib_pipe = {"api": "pislam_pipeline_create / _submit / _wait (include/pislam_hip.h): the timed steps ARE submits to this object",
                "depth": S, **pl.stats()}
    # dominant kernel alone: REP back-to-back launches inside one hipEvent bracket (a single eager launch is
    # bracketed together with ~10 us of command-processor latency)
    strip_ms = None
    if args.pipeline != 1:
        REP = 16
        ctx.set_option("repeat_strips", REP)
        try:
            rr = []
            for _ in range(3):
                fe1(d_pyr, kp, desc, counts)
                rr.append(fe1.last_timing()[1][0] / REP)
            strip_ms = float(np.mean(rr))
        finally:
            ctx.set_option("repeat_strips", 1)
    build_info = None
    if builder is not None:
        # what PISLAM_BUILD_MARGINS_CLEAN leaves out of the step: the margin pass of a build into a dirty buffer
        try:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            tm = []
            for clean in (True, False):
                with torch.cuda.stream(stream):
                    e0.record(stream)
                    for _ in range(10):
                        builder1(d_frames, d_pyr, margins_clean=clean)
                    e1.record(stream)
                torch.cuda.synchronize()
                tm.append(e0.elapsed_time(e1) / 10)
            build_info = {"build_ms_margins_clean": tm[0], "build_ms_with_margin_pass": tm[1],
                          "timed_step_uses": "margins_clean" if args.margins_clean else "margin pass every step"}
        except Exception:                                # noqa: BLE001
            pass

    match_info = None
    if m_out is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(5):
            matchHammingBatch(desc, counts, t_desc, t_counts, *m_out, ctx=ctx)
        e1.record(stream)
        torch.cuda.synchronize()
        cq = torch.clamp(counts.to(torch.int64), max=args.max_keypoints)
        ct = torch.clamp(t_counts.to(torch.int64), max=args.max_keypoints)
        pairs = int((cq * ct).sum().item())
        mm = e0.elapsed_time(e1) / 5
        match_info = {"ms": mm, "descriptor_pairs": pairs, "pairs_per_s": pairs / (mm * 1e-3),
                      "matched_within_64_bits": int(((m_out[1] <= 64) & (torch.arange(args.max_keypoints, device=dev)[None, :]
                                                                          < cq[:, None])).sum().item())}

    # (LAST of the measurements: its small, synchronised calls leave the GPU mostly idle — the strip-kernel bracket above
    #  would otherwise be taken on clocks that have dropped; round 5's first profile showed 0.173 ms there against
    #  0.161 ms in the kernel trace of the same kernel)
    if pl1 is not None:
        # ... and ONE pyramid per call on the same object (the reference's own use: a frame at a time)
        if not args.no_one_pyramid and b1 is None and m_out is None:
            try:
                o_s = fe.alloc_outputs(1, dev)
                d_one = d_pyr[:1]
                brackets = []
                with torch.cuda.stream(s1):
                    for _ in range(4):
                        pl1.submit(fe.params, fe.levels, d_one, *o_s)
                    for _ in range(3):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(s1)
                        for _ in range(200):
                            pl1.submit(fe.params, fe.levels, d_one, *o_s)
                        e1.record(s1)
                        brackets.append((e0, e1))
                torch.cuda.synchronize()
                one_pyr_ms = min(a.elapsed_time(b) for a, b in brackets) / 200
                if int(o_s[2].cpu()[0]) != int(o1[2].cpu()[0]):
                    raise RuntimeError("a pyramid alone and as the first of its batch gave different keypoint counts")
                # ... and the latency of ONE isolated call on an idle stream (events right around the submit, a synchronise
                # between calls): 200 back-to-back submits above are paced by the host's ~30 us per pislam_pipeline_submit
                # from Python, not by the GPU
                lat = []
                with torch.cuda.stream(s1):
                    for _ in range(60):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(s1)
                        pl1.submit(fe.params, fe.levels, d_one, *o_s)
                        e1.record(s1)
                        e1.synchronize()
                        lat.append(e0.elapsed_time(e1))
                one_pyr_lat_ms = float(np.median(lat[10:]))
                one_pyr_path = "one launch (pf::k_frame)" if (fe.last_path_of(c1) & 4) else "three launches"
            except Exception as e:                       # noqa: BLE001
                one_pyr_ms = None
                print(f"[bench] one-pyramid-per-call measurement failed: {e!r}", file=sys.stderr)

        pl1.close()
    deferred, nstrips = fe1.last_stats()
    # counts are the reference's un-clamped totals; keypoints beyond the capacity are neither stored nor
    # described, so only min(count, max_keypoints) per pyramid is credited
    capped = int((allc.to(torch.int64) > args.max_keypoints).sum().item())
    local_kp = int(torch.clamp(counts.to(torch.int64), max=args.max_keypoints).sum().item())    # lane 0's batch (side context)
    # keypoints of the K timed steps: step k ran lane k % S, whose batch (its own pyramids) yields the same counts every
    # time — every output set of a lane holds them
    lane_kp = [int(torch.clamp(P.outs[0][2].to(torch.int64), max=args.max_keypoints).sum().item()) for P in pipes]
    assert all(torch.equal(o[2], P.outs[0][2]) for P in pipes for o in P.outs), "a lane's output sets disagree"
    k_first = tick_timed[0]
    timed_kp = sum(lane_kp[k % S] for k in range(k_first, k_first + args.steps))                   # this rank
    if world > 1:
        import torch.distributed as dist
        tk = torch.tensor([timed_kp], dtype=torch.int64, device=dev if dist.get_backend() == "nccl" else None)
        dist.all_reduce(tk)
        timed_kp = int(tk.item())
    total_kp_step = timed_kp / args.steps                  # all ranks, mean over the timed steps
    value = timed_kp / dt

    parity_bad = False
    if rank == 0:
        fused = args.pipeline != 1
        valid_px = sum(t[0] * t[1] for t in levels)
        # algorithmic bytes of the DOMINANT KERNEL's launch: every valid pixel once + 36 B per
        # keypoint + 4 B count — the strip kernel reads no frame and writes no pyramid, so the 720p-build
        # workload's extra traffic (frame read + pyramid write) is priced with the whole step, never here
        b_alg_launch = B * (valid_px + 4) + 36 * local_kp
        b_alg_step = b_alg_launch + (B * (w0 * h0 + valid_px) if builder is not None else 0)
        # dominant kernel: k_fused_strips (one launch per step covers the whole batch); for the staged
        # pipeline there is no single dominant launch, so the whole step is priced instead
        launch_ms = (strip_ms if strip_ms else ev_stage_ms[0]) if fused else ev_total_ms
        achieved = b_alg_launch / (launch_ms * 1e-3) / 1e9
        step_ms = dt / args.steps * 1e3
        bsfx = f"_buckets{args.log_bucket_size}{args.bucket_limit}" if args.log_bucket_size else ""
        prof, prof_source = load_counters(args.workload, B, bsfx) if fused else (None, "null: staged pipeline")
        prof3, prof3_source = load_counters(args.workload, B, bsfx + "_streams3") if fused else (None, "null: staged pipeline")
        traffic = traffic_step = traffic_step3 = valu = lds_info = None
        if prof3:
            traffic_step3 = sum(k.get("hbm_bytes_per_launch", 0) * k.get("launches_per_step", 1) for k in prof3["kernels"].values()) or None
        if prof:
            ks = prof["kernels"]
            dom = ks.get("k_fused_strips", {})
            traffic = dom.get("hbm_bytes_per_launch")
            traffic_step = sum(k.get("hbm_bytes_per_launch", 0) * k.get("launches_per_step", 1) for k in ks.values()) or None
            if dom.get("SQ_LDS_IDX_ACTIVE") and dom.get("SQ_BUSY_CU_CYCLES"):
                # the LDS of the dominant kernel: cycles its arrays were active over the CUs' busy cycles, and the share of
                # those cycles that were bank-conflict replays (MI355X_MICROARCH.md, LDS section)
                lds_info = {"busy_frac": dom["SQ_LDS_IDX_ACTIVE"] / dom["SQ_BUSY_CU_CYCLES"],
                            "conflict_frac": dom.get("SQ_LDS_BANK_CONFLICT", 0.0) / dom["SQ_LDS_IDX_ACTIVE"],
                            "lds_insts": dom.get("SQ_INSTS_LDS"), "wait_inst_lds": dom.get("SQ_WAIT_INST_LDS"),
                            "per_phase": f"profiles/{PROFILE_TAG}_phase_ablation.txt", "source": prof_source}
            if dom.get("SQ_INSTS_VALU") and clock_ghz:
                nsimd = 4 * 256
                insts = dom["SQ_INSTS_VALU"]
                rate = insts / (launch_ms * 1e-3)                 # wave64 VALU instructions per second, chip-wide
                valu = {"insts": insts, "salu_insts": dom.get("SQ_INSTS_SALU"),
                        "issue_frac": insts * VALU_CYCLES_PER_WAVE_INST / (nsimd * clock_ghz * 1e9 * launch_ms * 1e-3),
                        "issue_frac_of_sustained": rate / VALU_SUSTAINED_PER_S,
                        "wave_insts_per_s": rate, "sustained_wave_insts_per_s": VALU_SUSTAINED_PER_S,
                        "cycles_per_wave_inst_assumed": VALU_CYCLES_PER_WAVE_INST,
                        "clock_ghz": clock_ghz,
                        "note": "wave64 VALU instructions of one strip-kernel launch / launch_ms.  issue_frac prices an instruction "
                                "at 4 cycles of its SIMD (1024 SIMDs x shader clock / 4): the rate tools/probes/valu_rate.hip "
                                "MEASURED on this part for dependent-free 32-bit integer ops is 528-538 G wave-instructions/s "
                                "chip-wide = 4.5 cycles — /opt/skills/guides/MI355X_MICROARCH.md quotes 2 cycles per wave64 op, which "
                                "the probe does not reach for this instruction mix.  issue_frac_of_sustained = the rate over that "
                                "measured 533 G/s: the share of the ACHIEVABLE integer issue rate this kernel uses — the binding "
                                "resource.  Clock measured in this run (s_memtime vs s_memrealtime under the steps' load)",
                        "source": prof_source}
        cfg_idx = {"vga": 2 if world > 1 else 1, "1280x960": 3, "1280x960-dense": 3, "720p-build": 4, "demo-photo": 0}[args.workload]
        out = {
            "metric": METRIC,
            "value": value, "unit": "kp+desc/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": step_ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "the reference's demo photo (fixture), replicated" if args.workload == "demo-photo" else "synthetic",
            "one_batch_ms": one_ms,
            "one_pyramid_ms": one_pyr_ms,
            "one_pyramid_latency_ms": one_pyr_lat_ms,
            "one_pyramid_path": one_pyr_path,
            "one_batch_value": (local_kp / (one_ms * 1e-3)) if one_ms else None,
            "config": {
                "workload": {"vga": f"batch={B} synthetic 640x480 pyramids per GPU, 8 levels x1.2 stacked (vstep=640, "
                                    "2210 rows)",
                             "1280x960": f"batch={B} synthetic 1280x960 pyramids per GPU (~2000 keypoints each: 148 shapes per frame), "
                                         "8 levels x1.2, packed layout (vstep=1280, 3768 rows, levels 4|5 and 6|7 side by side)",
                             "1280x960-dense": f"batch={B} synthetic 1280x960 pyramids per GPU at the VGA shape density (320 shapes per "
                                               "frame, ~4400 keypoints each — denser than BASELINE.json's ~2000), 8 levels x1.2, packed "
                                               "layout (vstep=1280, 3768 rows, levels 4|5 and 6|7 side by side)",
                             "demo-photo": f"batch={B} copies per lane of the reference's own demo pyramid (tests/golden/demo_pyramid.npz: "
                                           "the photo demo.cpp runs on, natural image content), 640x480, 8 levels x1.2 stacked "
                                           "(vstep=640, 2210 rows)",
                             "720p-build": f"batch={B} synthetic 1280x720 frames per GPU; gaussian5x5 + "
                                           "13/16,7/8,13/16,13/16,7/8,13/16,13/16 bilinear pyramid built on the GPU inside "
                                           f"the step (vstep={vstep}, {rows} rows"
                                           + (", the rebuild vouches for the zero margins of its own first fill: "
                                              "PISLAM_BUILD_MARGINS_CLEAN — margin pass cost in config.pyramid_build"
                                              if args.margins_clean else ", margins re-zeroed every step") + ")"}[args.workload] +
                            ", border=16, FAST threshold=20, Harris threshold=1<<15, " + ("no buckets" if not args.log_bucket_size else f"buckets <{args.log_bucket_size},{args.bucket_limit}>") + ", "
                            f"256-bit descriptors (BASELINE.json configs[{cfg_idx}]" + (f": {world} GPUs" if world > 1 else "") + ")",
                "baseline_config_index": cfg_idx,
                "batch_per_gpu": B, "global_batch": B * world, "distinct_pyramids_per_gpu": nd * nlane_in,
                "input_buffers_per_lane": "shared (one buffer read by every lane)" if args.shared_input else "distinct",
                "input": (f"{nlane_in} device-resident batch(es) of {B} per GPU, {nd} different pyramids each "
                          f"(seeds 0x5eed0000 + (rank*{S} + lane)*{B} + i); lane l submits batch l"),
                "keypoints_per_step_by_lane": lane_kp,
                "keypoints_per_pyramid": total_kp_step / (B * world),
                "pyramids_per_s": B * world * args.steps / dt,
                "parallelism": f"pyramid-shard x{world}, all-gather of counts" if world > 1 else "single GPU",
                "pipeline": "fused" if fused else "staged",
                "launch": "hipGraph replay" if use_graphs else "eager",
                "streams": S,
                "batches_in_flight": (f"{S}: step k is pislam_pipeline_submit to lane k % {S} of ONE library pipeline object (own "
                                      "HIP stream, context/workspace, outputs, hipGraph inside the library); each step is one whole "
                                      "batch, all K steps start and finish inside the timed region; one_batch_ms = the same "
                                      "through a pipeline of depth 1, one_pyramid_ms (--one-pyramid) = ONE pyramid per call through that object" if S > 1 else "1"),
                "strips_redone_by_overflow_pass": f"{deferred} of {nstrips}",
                "max_keypoints": args.max_keypoints, "pyramids_over_capacity": capped,
                "count_allgather": xchg.path,
                "count_allgathers_per_rank": coll, "ramp_groups": ramp_groups,
                "rccl_ranks": rccl_ranks,
                "communicators_per_rank": 1 if (hub is not None or (world > 1 and not gloo_mode)) else 0,
                "dist_fallbacks": fallbacks,
                "library_options": dict(extra_opts),
                "library_pipeline": lib_pipe,
                "ms_per_step_ranks": {"min": min(dts) / args.steps * 1e3, "max": max(dts) / args.steps * 1e3,
                                      "rank0": dt_rank / args.steps * 1e3},
                **({"pyramid_build": build_info} if build_info else {}),
                **({"match_inside_step": match_info} if match_info else {}),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": prof_source,
                "traffic_whole_step": traffic_step,
                "traffic_whole_step_3_lanes": traffic_step3, "traffic_3_lanes_source": prof3_source,
                "kernel": "pf::k_fused_strips (hipEvents on the launch stream around 16 back-to-back launches, / 16, measured after "
                          "the timed region with the other pipelines idle)" if fused
                          else "whole staged step (all launches)",
                "algorithmic_bytes_per_launch": b_alg_launch, "launch_ms": launch_ms,
                "whole_step": {"algorithmic_bytes": b_alg_step, "ms": step_ms,
                               "achieved": b_alg_step / (step_ms * 1e-3) / 1e9,
                               "frac": b_alg_step / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                               "note": "per GPU, every kernel of a step over ms_per_step" +
                                       (" (bytes incl. frame read + pyramid write of the build)" if builder is not None else "")},
                "valu": valu,
                "lds": lds_info,
                "shader_clock_ghz": clock_ghz,
                "step_gpu_ms": ev_total_ms,
                "stage_ms": {"detect+score+nms": ev_stage_ms[0], "overflow pass" if fused else "extract": ev_stage_ms[1],
                             "gather+orb" if fused else "orb": ev_stage_ms[2]},
            },
        }
        kept = []
        lb, bl = args.log_bucket_size, args.bucket_limit
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(host, levels, budget_s=args.cpu_seconds, log_bucket=lb, bucket_limit=bl,
                                               keep=NP, kept=kept)
        if snap is not None and not args.ablate:
            from oracle import orc
            for b in range(len(kept), min(NP, len(host))):      # (no CPU leg in this run: the oracle's results are computed here)
                k_, d_, _ = orc.pyramid4(host[b], levels, log_bucket=lb, bucket_limit=bl, cap=CPU_CAP)
                kept.append((k_, d_))
            kept_full = [kept[i % len(kept)] for i in range(min(NP, B))]     # (batch index i holds host pyramid i % nd)
            pins = None
            if args.workload == "demo-photo":                 # the reference's own counts on its demo input
                pins = 1754 if lb == 0 else 1315 if (lb, bl) == (4, 3) else None
            out["parity_in_run"] = parity_in_run(snap, kept_full, lane_hosts, levels, args.max_keypoints, lb, bl, pins=pins)
            parity_bad = not out["parity_in_run"]["ok"]
        # every other BASELINE configuration, briefly (children of this process; the headline above is complete by now)
        # (only in the full default line: development / profiling runs pass --no-cpu-baseline and get none of it)
        if (world == 1 and not args.no_other_workloads and not args.no_cpu_baseline and args.workload == "vga" and not args.ablate
                and not args.log_bucket_size and not args.match and args.pipeline != 1 and not args.opt and B == 256):
            pl.synchronize()
            torch.cuda.synchronize()
            t_o = time.perf_counter()
            recs, others_ok = other_workloads(args)
            out["other_workloads"] = recs
            out["other_workloads_wall_s"] = round(time.perf_counter() - t_o, 1)
            if not others_ok:
                parity_bad = True
                print(f"[bench] other_workloads: a child failed its parity check: {recs}", file=sys.stderr, flush=True)
        print(json.dumps(out), flush=True)
        if parity_bad:
            print(f"[bench] parity_in_run FAILED: {out['parity_in_run'].get('first_mismatch')}", file=sys.stderr, flush=True)
    if world > 1:
        torch.distributed.barrier()
        try:
            ctx.dist_finalize()                          # our RCCL communicator first, while every rank is still alive
        except Exception:                                # noqa: BLE001
            pass
        pl.synchronize()
        torch.distributed.destroy_process_group()
    return 6 if parity_bad else 0


def main():
    ap = build_parser()
    args = ap.parse_args()
    if not args.max_keypoints:
        args.max_keypoints = 4096 if args.workload == "vga" else 8192
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    argv = sys.argv[1:]
    if args.worker:
        with Watchdog(max(30.0, args.attempt_timeout + 30), "the worker", rc=5):
            return worker_main(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU under
        # torch.distributed.run on 127.0.0.1); rank 0 prints the JSON line.
        import torch
        from pislam_amd import dist as pdist
        shared_ok = args.dist_backend == "gloo" or args.selftest_spawn     # test modes: ranks may share a GPU / need none
        if not shared_ok and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible "
                             "(one process per GPU; there is no CPU fallback)")
        return pdist.self_launch([os.path.abspath(__file__)] + argv, args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                         "(or run `python bench.py --gpus N` without a torchrun environment)")
    if world > 1:
        return supervisor_main(args, argv)
    with Watchdog(args.wall_cap, "bench.py", rc=5):
        return worker_main(args)


if __name__ == "__main__":
    sys.exit(main() or 0)
