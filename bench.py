#!/usr/bin/env python3
"""bench.py — ORB keypoints+descriptors/sec on synthetic 640x480 8-level x1.2 pyramids.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the whole hot path (fastDetect -> fastScoreHarris -> fastExtract ->
orbCompute on every level of every pyramid) over one device-resident batch of 256 synthetic
pyramids per GPU (BASELINE.json configs[1]); for N>1 each rank owns its own 256 pyramids (weak
scaling) and the step ends with the RCCL all-gather of the per-pyramid keypoint counts.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(pyr_host, levels, budget_s=10.0, mt_s=3.0):
    """The oracle (bit-exact plain-C restatement of the reference path, oracle/pislam_oracle.c) timed
    on ONE host thread — the reference itself is single-threaded — on a bounded sample of the same
    workload; then the same port on EVERY visible host thread (pthreads inside liborc, one pyramid per
    thread at a time, timed in C) for >= mt_s seconds.  levels: (w, h, row0[, col0])."""
    from oracle import orc
    orc.lib()
    CPU_CAP = 16384                                       # output capacity per pyramid (keeps allocation out of the timing)
    orc.pyramid4(pyr_host[0], levels, cap=CPU_CAP)        # warm caches / lazy table
    n_kp, n_pyr, t0 = 0, 0, time.perf_counter()
    for b in range(64 * len(pyr_host)):                   # ~budget_s of work: the batch, repeated if need be
        kp, _, _ = orc.pyramid4(pyr_host[b % len(pyr_host)], levels, cap=CPU_CAP)
        n_kp += len(kp)
        n_pyr += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    out = {"value": n_kp / dt, "unit": "kp+desc/s", "cores": 1, "kind": "port",
           "sample": f"{n_pyr} pyramids of the batch ({n_kp} keypoints) in {dt:.2f} s, 1 thread, "
                     f"oracle/pislam_oracle.c -O3 on {os.cpu_count()} visible host cores"}
    # reported beside the single-thread figure, which stays `value`
    try:
        nthr = max(1, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
        tot, done, dt2 = orc.pyramid_mt(np.ascontiguousarray(pyr_host), levels, nthr, mt_s, cap=CPU_CAP)
        out["all_threads"] = {"value": tot / dt2, "cores": nthr,
                              "sample": f"{done} pyramids ({tot} keypoints) in {dt2:.2f} s, {nthr} pthreads (orc_pyramid_mt)"}
    except Exception as e:                                   # never let the extra leg break the bench line
        out["all_threads"] = {"error": repr(e)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--spin-s", type=float, default=0.5,
                    help="seconds of untimed load before the warm-up steps (GPU clock ramp)")
    ap.add_argument("--batch", type=int, default=256, help="pyramids per GPU")
    ap.add_argument("--distinct", type=int, default=0,
                    help="distinct synthetic pyramids generated per GPU (0 = all of the batch)")
    ap.add_argument("--max-keypoints", type=int, default=0,
                    help="keypoint / descriptor capacity per pyramid (0 = 4096 for vga, 8192 for the larger workloads)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="testing only (N=1): run the per-step count all-gather anyway, through a 1-rank RCCL communicator "
                         "created by the C ABI (pislam_dist_*), so that the N>1 data path is exercised on a 1-GPU box")
    ap.add_argument("--selftest-spawn", action="store_true",
                    help="testing only: exercise launch + rendezvous + count exchange with fake counts on the CPU "
                         "(no GPU work, value is null)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the 1-thread cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default=None,
                    help="override the torch.distributed backend (default nccl = RCCL); 'gloo' lets the N>1 code "
                         "path be exercised with several ranks sharing one GPU (testing only)")
    ap.add_argument("--workload", default="vga", choices=["vga", "1280x960", "720p-build"],
                    help="vga = BASELINE configs[1] (default, the headline); 1280x960 = configs[3] (packed layout, "
                         "vstep 1280); 720p-build = configs[4] (gaussian5x5 + bilinear pyramid build on the GPU inside "
                         "the timed step)")
    ap.add_argument("--pipeline", type=int, default=0, help="0 auto (fused), 1 staged, 2 fused")
    ap.add_argument("--strip-rows", type=int, default=0)
    ap.add_argument("--orb-chunks", type=int, default=0)
    ap.add_argument("--wgs-per-cu", type=int, default=0)
    ap.add_argument("--xtile-cols", type=int, default=-1)
    ap.add_argument("--run-len", type=int, default=0)
    ap.add_argument("--alias", type=int, default=-1)
    ap.add_argument("--run-order", type=int, default=-1, help="1 (default): a pyramid's runs are launched longest first; 0: entry order")
    ap.add_argument("--strip-px", type=int, default=0, help="profiling: pixels per strip the height heuristic aims at (default 16384)")
    ap.add_argument("--strip-rows-max", type=int, default=0, help="profiling: upper bound of the heuristic strip height (default: 56 for large launches, 28 for small ones)")
    ap.add_argument("--tile-cols", type=int, default=0, help="levels with more classified columns are cut into x-tiles (0 = default 704, < 0 never)")
    ap.add_argument("--orb-in-strip", type=int, default=-1, help="1: strips describe their own keypoints; 0 (default): one gather+ORB pass")
    ap.add_argument("--graph", type=int, default=1,
                    help="1 (default): capture the step's launches into a hipGraph (torch.cuda.CUDAGraph) per output set "
                         "and replay it — falls back to eager launches if the capture or its check fails; 0: eager")
    ap.add_argument("--streams", type=int, default=3,
                    help="independent pipelines (HIP stream + context + outputs + graph); step k runs on pipeline "
                         "k %% S, so consecutive batches overlap on the GPU (default 3; 1 = strictly one batch at a time)")
    ap.add_argument("--match", action="store_true",
                    help="also match every pyramid's descriptors against its neighbour's inside the step (SURVEY 8f-4)")
    ap.add_argument("--match-mfma", type=int, default=-1, help="--match: 1 (default) matrix-core matcher, 0 the VALU popcount kernel")
    ap.add_argument("--lds-pad", type=int, default=0, help="profiling only: extra LDS per strip workgroup")
    ap.add_argument("--log-bucket-size", type=int, default=0, help="fastExtract logBucketSize (README uses 4)")
    ap.add_argument("--bucket-limit", type=int, default=5, help="fastExtract bucketLimit (README uses 3)")
    ap.add_argument("--ablate", type=int, default=0, help="profiling only (results invalid): fused-kernel phase mask")
    args = ap.parse_args()
    if not args.max_keypoints:
        args.max_keypoints = 4096 if args.workload == "vga" else 8192

    from pislam_amd import dist as pdist

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU under
        # torch.distributed.run on 127.0.0.1); rank 0 prints the JSON line.
        shared_ok = args.dist_backend == "gloo" or args.selftest_spawn     # test modes: ranks may share a GPU / need none
        if not shared_ok and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible "
                             "(one process per GPU; there is no CPU fallback)")
        raise SystemExit(pdist.self_launch([os.path.abspath(__file__)] + sys.argv[1:], args.gpus))

    rank, local_rank, world = pdist.init(backend="gloo" if args.selftest_spawn else args.dist_backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                         "(or run `python bench.py --gpus N` without a torchrun environment)")
    if args.selftest_spawn:
        # launch / rendezvous / exchange plumbing only (CPU, gloo): every rank contributes fake counts
        B = 4
        fake = torch.arange(B, dtype=torch.int32) + 100 * rank
        xchg = pdist.CountExchange(world)
        xchg.before_step()
        xchg.start(fake)
        allc = xchg.finish()
        ok = allc.tolist() == [100 * r + i for r in range(world) for i in range(B)]
        if world > 1:
            torch.distributed.barrier()
        if rank == 0:
            print(json.dumps({"metric": "ORB keypoints+descriptors/sec, 640x480 8-level pyramid", "value": None,
                              "unit": "kp+desc/s", "n_gpus": world, "selftest": "spawn", "exchange_ok": ok,
                              "count_allgather": xchg.path}))
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    from pislam_amd import synth
    from pislam_amd.frontend import OrbFrontend
    from pislam_amd.capi import Context

    if args.dist_backend == "gloo":
        local_rank = local_rank % max(1, torch.cuda.device_count())   # ranks may share a GPU in this test mode
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    builder = None
    d_frames = None
    if args.workload == "vga":
        levels = synth.level_table()                       # demo.cpp:38-47 table, 2210 stacked rows
        vstep, w0, h0 = 640, 640, 480
    elif args.workload == "1280x960":
        levels = synth.packed_level_table(1280, 960)       # 3768 rows: levels 4|5 and 6|7 side by side
        vstep, w0, h0 = 1280, 1280, 960
    else:
        from pislam_amd.frontend import PyramidBuilder
        w0, h0 = 1280, 720
    B = args.batch
    distinct = args.distinct or (B if args.workload == "vga" else min(B, 16))
    first = rank * B
    S = 1 if args.match else max(1, args.streams)
    force = args.force_exchange and world == 1
    # Output sets per pipeline: two whenever counts are exchanged, so that a pipeline's next step (into the other
    # set) does not wait for the all-gather of its previous one — the collective's latency (event hand-over to the
    # collective stream + a latency-bound RCCL kernel, ~50 us) would otherwise be added to every pipeline cycle
    # (measured on a 1-rank communicator, 3 pipelines: 0.275 ms per step with one set).
    nsets = 2 if (world > 1 or force) else 1

    def make_context(stream):
        c = Context(device=local_rank, stream=stream.cuda_stream)
        c.set_option("pipeline", args.pipeline)
        c.set_option("strip_rows", args.strip_rows)
        c.set_option("orb_chunks", args.orb_chunks)
        c.set_option("lds_pad", args.lds_pad)
        if args.alias >= 0:
            c.set_option("alias", args.alias)
        if args.run_order >= 0:
            c.set_option("run_order", args.run_order)
        if args.strip_px:
            c.set_option("strip_px", args.strip_px)
        if args.strip_rows_max:
            c.set_option("strip_rows_max", args.strip_rows_max)
        if args.tile_cols:
            c.set_option("tile_cols", args.tile_cols)
        if args.orb_in_strip >= 0:
            c.set_option("orb_in_strip", args.orb_in_strip)
        if args.match_mfma >= 0:
            c.set_option("match_mfma", args.match_mfma)
        if args.run_len:
            c.set_option("run_len", args.run_len)
        if args.xtile_cols >= 0:
            c.set_option("xtile_cols", args.xtile_cols)
        if args.wgs_per_cu:
            c.set_option("wgs_per_cu", args.wgs_per_cu)
        c.set_option("ablate", args.ablate)
        return c

    # ---- the resident input (shared by every pipeline; the 720p workload builds its pyramids per pipeline) ----
    host = d_frames = d_pyr0 = None
    if args.workload == "720p-build":
        fr = np.stack([synth.make_level0(first + i, w0, h0) for i in range(min(distinct, B))])
        d_frames = torch.from_numpy(fr).to(dev)
        if distinct < B:
            d_frames = d_frames[torch.arange(B, device=dev) % distinct].contiguous()
    else:
        rows = synth.pyramid_rows(levels)
        host = synth.make_batch(first, min(distinct, B), w0=w0, h0=h0, vstep=vstep, levels=levels)
        d_pyr0 = torch.from_numpy(host).to(dev)
        if distinct < B:
            d_pyr0 = d_pyr0[torch.arange(B, device=dev) % distinct].contiguous()

    # ---- S independent pipelines: batch k runs on pipeline k % S — its own HIP stream, context (workspace),
    # outputs and hipGraph — so that the gather+ORB kernel of one batch (latency / LDS bound) and the tail of its
    # strip kernel run under the strip kernel of the next batch (VALU bound).  Every batch is still processed
    # completely (strips -> overflow pass -> gather+ORB, then the count all-gather) inside the timed region.
    class Pipe:
        pass

    pipes = []
    rccl_note = None
    for i in range(S):
        P = Pipe()
        P.stream = torch.cuda.Stream(dev)
        P.ctx = make_context(P.stream)
        P.builder = None
        with torch.cuda.stream(P.stream):
            if args.workload == "720p-build":
                P.builder = PyramidBuilder(w0, h0, ctx=P.ctx)
                levels, vstep, rows = P.builder.levels, P.builder.vstep, P.builder.rows
                P.d_pyr = torch.empty((B, rows, vstep), dtype=torch.uint8, device=dev)
                P.builder(d_frames, P.d_pyr)
                torch.cuda.synchronize()
                if host is None:
                    host = P.d_pyr[:min(distinct, B)].cpu().numpy()
            else:
                P.d_pyr = d_pyr0
            P.fe = OrbFrontend(levels, vstep=vstep, rows=rows, max_keypoints=args.max_keypoints, ctx=P.ctx,
                               log_bucket_size=args.log_bucket_size, bucket_limit=args.bucket_limit)
            P.fe.reserve(B)
            P.outs = [P.fe.alloc_outputs(B, dev) for _ in range(nsets)]
        # The count all-gather goes through the C ABI (pislam_dist_*: RCCL communicator from a unique id,
        # ncclAllGather on the context's collective stream), one communicator per pipeline context.  Ranks
        # sharing one GPU (gloo test mode) cannot form an RCCL communicator; if the C-ABI path fails on any
        # rank, all ranks fall back to torch.distributed's all-gather and the JSON line says so.
        err = None
        if world > 1 and args.dist_backend == "gloo":
            err = "test mode: ranks share a GPU"
        elif world > 1:
            err = pdist.init_rccl(P.ctx, rank, world, dev)
            if err and rank == 0:
                print(f"[bench] C-ABI RCCL path unavailable, using torch.distributed: {err}", file=sys.stderr)
        if force:
            from pislam_amd import capi
            P.ctx.set_option("dist_rccl_single", 1)
            P.ctx.dist_init(capi.dist_unique_id(), 0, 1)
        P.xchg = pdist.CountExchange(world, ctx=P.ctx if ((world > 1 and err is None) or force) else None,
                                     always_collective=force, sets=nsets)
        rccl_note = rccl_note or err
        pipes.append(P)
    ctx, fe, builder, d_pyr = pipes[0].ctx, pipes[0].fe, pipes[0].builder, pipes[0].d_pyr
    stream = pipes[0].stream
    kp, desc, counts = pipes[0].outs[0]
    xchg = pipes[0].xchg

    m_out = None
    if args.match:
        # train side: the neighbouring pyramid's descriptors (static inputs -> made once, outside the step)
        from pislam_amd.frontend import matchHammingBatch
        with torch.cuda.stream(stream):
            fe(d_pyr, kp, desc, counts)
        torch.cuda.synchronize()
        t_desc, t_counts = torch.roll(desc, 1, 0).contiguous(), torch.roll(counts, 1, 0).contiguous()
        m_out = [torch.empty((B, args.max_keypoints), dtype=torch.int32, device=dev) for _ in range(3)]
    nstep = [0]

    def launches(P, k_, d_, c_):
        if P.builder is not None:
            # steady state of a stream: the same pyramid buffer is refilled every step by the same builder
            # (its first fill, before the timed region, established the zero margins)
            P.builder(d_frames, P.d_pyr, margins_clean=True)
        P.fe(P.d_pyr, k_, d_, c_)
        if m_out is not None:
            matchHammingBatch(d_, c_, t_desc, t_counts, *m_out, ctx=P.ctx)

    use_graphs = False
    if args.graph:
        # The batch call allocates nothing and never synchronises once the workspace is reserved, so a step's
        # launches can be replayed from a hipGraph.  Any failure (capture error, replay not reproducing the
        # eager counts) falls back to eager launches: the measurement must never depend on this.
        try:
            for P in pipes:
                with torch.cuda.stream(P.stream):
                    for o in P.outs:
                        launches(P, *o)                 # warm-up outside the capture (allocations, module load)
            torch.cuda.synchronize()
            for P in pipes:
                want = [o[2].clone() for o in P.outs]
                P.graphs = []
                for o in P.outs:
                    g = torch.cuda.CUDAGraph()
                    # thread_local: API calls of other threads (the RCCL watchdog of a multi-rank run) must not
                    # invalidate the capture
                    with torch.cuda.graph(g, stream=P.stream, capture_error_mode="thread_local"):
                        launches(P, *o)
                    P.graphs.append(g)
                for o, g, w in zip(P.outs, P.graphs, want):
                    o[2].zero_()
                    torch.cuda.synchronize()
                    with torch.cuda.stream(P.stream):
                        g.replay()
                    torch.cuda.synchronize()
                    if not torch.equal(o[2], w):
                        raise RuntimeError("graph replay does not reproduce the eager result")
            use_graphs = True
        except Exception as e:                          # noqa: BLE001
            print(f"[bench] hipGraph path disabled: {e!r}", file=sys.stderr)
            torch.cuda.synchronize()
    graphs = pipes[0].graphs if use_graphs else None

    def step():
        k = nstep[0]
        nstep[0] += 1
        P = pipes[k % S]
        i = (k // S) % nsets
        with torch.cuda.stream(P.stream):
            P.xchg.before_step()                        # this pipeline's stream waits for the all-gather that read set i
            if use_graphs:
                P.graphs[i].replay()
            else:
                launches(P, *P.outs[i])
            P.xchg.start(P.outs[i][2])
        return P

    def spin_once():
        for P in pipes:
            with torch.cuda.stream(P.stream):
                if use_graphs:
                    P.graphs[0].replay()
                else:
                    launches(P, *P.outs[0])

    # Clock ramp: the GPU idles at a few hundred MHz and needs a fraction of a second of load to reach its
    # sustained clocks — far longer than a handful of 0.4 ms steps.  Spin the same step, untimed, before the W
    # warm-up steps so that W and K measure the steady state whatever their values.
    # (No collective in here: the loop is time-based, so ranks run different iteration counts.)
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < args.spin_s:
        for _ in range(8):
            spin_once()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    for P in pipes:
        P.xchg.finish()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    nstep[0] = 0
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
    allc = None
    for P in pipes:                                     # every step's all-gather has completed
        r = P.xchg.finish()
        if P is last:
            allc = r
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if torch.distributed.get_backend() == "nccl" else None)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    # device time of the library's launches for the last step (hipEvents on the launch stream)
    ev_total_ms, ev_stage_ms = fe.last_timing()
    # a few more event-timed steps for an average kernel-side duration
    ev = []
    for _ in range(min(5, args.steps)):
        fe(d_pyr, kp, desc, counts)
        ev.append(fe.last_timing())
    ev_total_ms = float(np.mean([e[0] for e in ev]))
    ev_stage_ms = [float(np.mean([e[1][i] for e in ev])) for i in range(3)]
    # dominant kernel alone: REP back-to-back launches inside one hipEvent bracket (a single eager launch is
    # bracketed together with ~10 us of command-processor latency)
    strip_ms = None
    if args.pipeline != 1:
        REP = 16
        ctx.set_option("repeat_strips", REP)
        try:
            rr = []
            for _ in range(3):
                fe(d_pyr, kp, desc, counts)
                rr.append(fe.last_timing()[1][0] / REP)
            strip_ms = float(np.mean(rr))
        finally:
            ctx.set_option("repeat_strips", 1)

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

    deferred, nstrips = fe.last_stats()
    # counts are the reference's un-clamped totals; keypoints beyond the capacity are neither stored nor
    # described, so only min(count, max_keypoints) per pyramid is credited
    capped = int((allc.to(torch.int64) > args.max_keypoints).sum().item())
    total_kp_step = int(torch.clamp(allc.to(torch.int64), max=args.max_keypoints).sum().item())   # all ranks, one step
    local_kp = int(torch.clamp(counts.to(torch.int64), max=args.max_keypoints).sum().item())
    value = total_kp_step * args.steps / dt

    if rank == 0:
        valid_px = sum(t[0] * t[1] for t in levels)
        if builder is not None:            # config 5: + source frame read + pyramid write
            valid_px += w0 * h0 + sum(t[0] * t[1] for t in levels)
        # algorithmic bytes: every valid pixel once + 36 B per keypoint + 4 B count
        b_alg_launch = B * (valid_px + 4) + 36 * local_kp
        fused = args.pipeline != 1
        # dominant kernel: k_fused_strips (one launch per step covers the whole batch); for the staged
        # pipeline there is no single dominant launch, so the whole step is priced instead
        launch_ms = (strip_ms if strip_ms else ev_stage_ms[0]) if fused else ev_total_ms
        achieved = b_alg_launch / (launch_ms * 1e-3) / 1e9
        # HBM bytes per launch come from separate rocprofv3 --pmc passes (tools/pmc_hbm.sh), which cannot run
        # inside this process: the number is read from the committed profile and reported ONLY when that
        # profile was taken on exactly these kernel sources and this workload; `traffic_source` says where it
        # is from (or why it is null).
        from pislam_amd import build as pbuild
        traffic, traffic_step, traffic_source = None, None, None
        tpath = os.path.join(ROOT, "profiles", "r02_hbm_traffic.json")
        if fused and B == 256 and args.workload == "vga" and not args.log_bucket_size and os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("source_hash") == pbuild.source_hash():
                    traffic = tj["kernels"]["k_fused_strips"]["hbm_bytes_per_launch"]
                    traffic_step = sum(k["hbm_bytes_per_launch"] for k in tj["kernels"].values() if "hbm_bytes_per_launch" in k)
                    traffic_source = f"profiles/r02_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes on kernel sources {tj['source_hash']}; not measured in this run)"
                else:
                    traffic_source = (f"null: profiles/r02_hbm_traffic.json was measured on kernel sources {tj.get('source_hash')}, "
                                      f"this run uses {pbuild.source_hash()}")
            except Exception as e:                          # noqa: BLE001
                traffic_source = f"null: {e!r}"
        else:
            traffic_source = "null: no PMC profile for this workload / configuration"
        out = {
            "metric": "ORB keypoints+descriptors/sec, 640x480 8-level pyramid",
            "value": value, "unit": "kp+desc/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {
                "workload": {"vga": f"batch={B} synthetic 640x480 pyramids per GPU, 8 levels x1.2 stacked (vstep=640, "
                                    "2210 rows)",
                             "1280x960": f"batch={B} synthetic 1280x960 pyramids per GPU, 8 levels x1.2, packed layout "
                                         "(vstep=1280, 3768 rows, levels 4|5 and 6|7 side by side)",
                             "720p-build": f"batch={B} synthetic 1280x720 frames per GPU; gaussian5x5 + "
                                           "13/16,7/8,13/16,13/16,7/8,13/16,13/16 bilinear pyramid built on the GPU inside "
                                           f"the step (vstep={vstep}, {rows} rows)"}[args.workload] +
                            ", border=16, FAST threshold=20, Harris threshold=1<<15, " + ("no buckets" if not args.log_bucket_size else f"buckets <{args.log_bucket_size},{args.bucket_limit}>") + ", "
                            "256-bit descriptors (BASELINE.json configs[1])",
                "batch_per_gpu": B, "global_batch": B * world, "distinct_pyramids_per_gpu": min(distinct, B),
                "keypoints_per_pyramid": total_kp_step / (B * world),
                "pyramids_per_s": B * world * args.steps / dt,
                "parallelism": f"pyramid-shard x{world}, RCCL all-gather of counts" if world > 1 else "single GPU",
                "pipeline": "fused" if fused else "staged",
                "launch": "hipGraph replay" if graphs is not None else "eager",
                "streams": S,
                "batches_in_flight": f"{S}: step k runs on pipeline k % {S} (own HIP stream, context/workspace, outputs, "
                                     "graph); each step is one whole batch, all K steps start and finish inside the "
                                     "timed region" if S > 1 else "1",
                "strips_redone_by_overflow_pass": f"{deferred} of {nstrips}",
                "max_keypoints": args.max_keypoints, "pyramids_over_capacity": capped,
                "count_allgather": xchg.path,
                **({"match_inside_step": match_info} if match_info else {}),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                "traffic_whole_step": traffic_step,
                "kernel": "pf::k_fused_strips (hipEvents on the launch stream around 16 back-to-back launches, / 16, measured after "
                          "the timed region with the other pipelines idle)" if fused
                          else "whole staged step (all launches)",
                "algorithmic_bytes_per_launch": b_alg_launch, "launch_ms": launch_ms,
                "step_gpu_ms": ev_total_ms,
                "stage_ms": {"detect+score+nms": ev_stage_ms[0], "overflow pass" if fused else "extract": ev_stage_ms[1],
                             "gather+orb" if fused else "orb": ev_stage_ms[2]},
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(host, levels, budget_s=args.cpu_seconds)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        for P in pipes:
            try:
                P.ctx.dist_finalize()                    # our RCCL communicators first, while every rank is still alive
            except Exception:                            # noqa: BLE001
                pass
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
