#!/bin/bash
# usage (GPU box): bash tools/probes/frame_sweep.sh [bench args]  — the one-launch path (pf::k_frame) at ONE pyramid per call under
# different strip plans: kernel time (rocprofv3 --kernel-trace --stats) and call-to-call time (bench.py --batch 1 --streams 1)
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
run() {
  d=$root/gpurun_out/fsw_$$
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python $root/bench.py --batch 1 --streams 1 --steps 400 --warmup 40 --no-cpu-baseline --no-one-pyramid --parity-pyramids 1 "$@" > $d.json 2>/dev/null
  python - <<P
import csv, json
try:
    d = json.loads(open("$d.json").read().strip().splitlines()[-1])
    ms, kp, ok = d["ms_per_step"], d["config"]["keypoints_per_pyramid"], (d.get("parity_in_run") or {}).get("ok")
except Exception as e:
    ms, kp, ok = None, None, repr(e)
rows = [r for r in csv.DictReader(open("$d/p_kernel_stats.csv")) if int(r["Calls"]) > 100]
print("$*", "| call ms", ms, "kp", kp, "parity", ok, "|", [(r["Name"].split("(")[0][-22:], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2)) for r in rows[:3]])
P
  rm -rf $d $d.json
}
for w in ""; do
run $w "$@"
run $w --opt strip_rows_max=16 "$@"
run $w --opt strip_rows_max=20 "$@"
run $w --opt tile_cols=320 "$@"
run $w --opt tile_cols=320 --opt strip_rows_max=16 "$@"
run $w --opt tile_cols=224 --opt strip_rows_max=16 "$@"
run $w --opt tile_cols=224 --opt strip_rows_max=20 "$@"
run $w --opt tile_cols=160 --opt strip_rows_max=16 "$@"
run $w --opt tile_cols=160 --opt strip_rows_max=24 "$@"
run $w --opt tile_cols=320 --opt strip_rows_max=16 --opt orb_chunks=32 "$@"
run $w --opt tile_cols=320 --opt strip_rows_max=16 --opt orb_chunks=128 "$@"
run $w --opt frame=0 "$@"
done
