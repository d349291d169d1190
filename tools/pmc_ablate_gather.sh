#!/bin/bash
# per-phase instruction counts AND LDS cycles of pf::k_gather_orb: PMC passes over its profiling instantiation with cumulative
# ablations of the describe loop (option "ablate" bits 20..23, pf::orb_describe `glevel`) — the difference of two lines is what
# the phase between them costs:
#   1 = prologue only (scan of the strip counts, per-keypoint strip search, keypoint copy), 2 = + patch fetch (global loads),
#   3 = + parking the windows in LDS, 4 = + row read-back, moments and the two 32-lane sums, 5 = + angle bin,
#   6 = + BRIEF offset-table loads, 7 = + BRIEF sample reads and bit assembly, 8 = everything (+ descriptor store); 0 = product.
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
for g in 1 2 3 4 5 6 7 8 0; do
  a=$((g << 20))
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $root/gpurun_out/gabl_$g -o p -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-one-pyramid --no-other-workloads --parity-pyramids 0 --graph 0 --streams 1 --ablate $a "$@" > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/gablt_$g -o p -- python $root/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-one-pyramid --no-other-workloads --parity-pyramids 0 --graph 0 --streams 1 --ablate $a "$@" > /dev/null 2>&1
  python - <<P
import csv, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open("$root/gpurun_out/gabl_$g/p_counter_collection.csv")):
    if "k_gather_orb" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
us = [float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open("$root/gpurun_out/gablt_$g/p_kernel_stats.csv")) if "k_gather_orb" in r["Name"]]
print("gather level $g", {c: round(sorted(v)[len(v)//2] / 1e6, 2) for c, v in sorted(agg.items())}, "kernel us", [round(u, 1) for u in us])
P
  rm -rf $root/gpurun_out/gabl_$g $root/gpurun_out/gablt_$g
done
