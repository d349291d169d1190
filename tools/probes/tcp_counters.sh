#!/bin/bash
# usage (GPU box): bash tools/probes/tcp_counters.sh [bench args]  — vector-L1 (TCP) / L2 request counters of the pf:: kernels,
# one batch at a time: how many line requests the gather's patch fetch sends to the L2 and how long they take
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
for set in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TCC_REQ_sum TCC_READ_sum"; do
  d=$root/gpurun_out/tcpq_$$
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -o p -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-one-pyramid --no-other-workloads --parity-pyramids 0 --graph 0 --streams 1 "$@" > /dev/null 2>&1
  python - <<P
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
try:
    for r in csv.DictReader(open("$d/p_counter_collection.csv")):
        if "pf::" not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, dd in agg.items():
        print("$*", k, {c: round(sorted(v)[len(v)//2] / 1e6, 3) for c, v in dd.items()})
except Exception as e:
    print("no counters for set: $set", e)
P
  rm -rf $d
done
