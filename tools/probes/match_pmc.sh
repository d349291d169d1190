#!/bin/bash
# SQ counters of the stand-alone matcher probe (tools/probes/_bin/match_probe_0)
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM"; do
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $root/gpurun_out/mpmc -o p -- $root/tools/probes/_bin/match_probe_0 > /dev/null 2>&1
  python - <<P
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$root/gpurun_out/mpmc/p_counter_collection.csv")):
    agg[r["Kernel_Name"].split("(")[0][-20:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sorted(v)[len(v)//2] / 1e6, 2) for c, v in d.items()})
P
  rm -rf $root/gpurun_out/mpmc
done
