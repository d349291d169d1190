#!/bin/bash
# per-phase instruction counts AND LDS cycles (active, bank conflicts, waiting to issue) of the strip kernel: PMC passes over
# the profiling build with cumulative ablations — the difference of two lines is what the phase between them costs
# (1 = stop after staging, 16 = prefilter only, 2 = + pretest, 4 = + FAST-9, 8 = + Harris, 65536 = everything)
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
for a in 1 16 2 4 8 65536 0; do   # (0 = the product kernel)
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $root/gpurun_out/abl_$a -o p -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-one-pyramid --parity-pyramids 0 --graph 0 --ablate $a "$@" > /dev/null 2>&1
  python - <<P
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$root/gpurun_out/abl_$a/p_counter_collection.csv")):
    k = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
    if k in ("k_fused_strips",): agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print("ablate $a", k, {c: round(sorted(v)[len(v)//2] / 1e6, 2) for c, v in d.items()})
P
done
