#!/bin/bash
# usage (GPU box): bash tools/pmc_quick.sh <bench args...>  — SQ instruction counters of the pf:: kernels (M per launch,
# median over the launches) for one bench configuration, one batch at a time
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
d=$root/gpurun_out/pmcq_$$
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES --output-format csv -d $d -o p -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-one-pyramid --parity-pyramids 0 --graph 0 --streams 1 "$@" > /dev/null 2>&1
python - <<P
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$d/p_counter_collection.csv")):
    if "pf::" not in r["Kernel_Name"]: continue
    k = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, dd in agg.items():
    print("$*", k, {c: round(sorted(v)[len(v)//2] / 1e6, 2) for c, v in dd.items()})
P
rm -rf $d
