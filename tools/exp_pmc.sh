#!/bin/bash
# development: which SQ / LDS / TA counters exist, and a wider counter set for the pf:: kernels (one batch at a time)
out=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/exppmc; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --list-avail > $out/avail.txt 2>&1
grep -o "SQ_[A-Z0-9_]*\|TA_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|LDS[A-Za-z0-9_]*" $out/avail.txt | sort -u > $out/avail_names.txt
i=0
for pmc in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_WAVES SQ_BUSY_CU_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $out/p$i -o p -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --graph 0 --streams 1 --spin-s 0.1 > /dev/null 2> $out/err$i.txt
  i=$((i+1))
done
python - <<P
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/p*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "pf::" not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, dd in agg.items():
    print(k, {c: round(sorted(v)[len(v)//2] / 1e6, 3) for c, v in sorted(dd.items())})
P
rm -rf $out/p?
