#!/bin/bash
# kernel-trace of the matcher inside bench.py --match (durations of k_match_mfma / k_match)
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
for m in 1 0; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/mt_$m -o p -- python $root/bench.py --steps 5 --warmup 1 --no-cpu-baseline --match --match-mfma $m > /dev/null 2>&1
  grep "k_match" $root/gpurun_out/mt_$m/p_kernel_stats.csv | cut -c1-60,200-400
  rm -rf $root/gpurun_out/mt_$m
done
