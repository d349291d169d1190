#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>   e.g. r02
# Produces under gpurun_out/<tag>/ everything profiles/ keeps for a round: bench lines of the three workloads,
# rocprofv3 --kernel-trace --stats of the same commands, HBM traffic (separate --pmc passes) and SQ counters.
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# HBM traffic first (separate --pmc passes), so that the bench lines below can report it: bench.py reads
# profiles/<tag>_hbm_traffic.json and uses it only when its source_hash matches the kernels it runs
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $root/gpurun_out/${tag}hbm_fetch -o p -- python $root/bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $root/gpurun_out/${tag}hbm_write -o p -- python $root/bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $root/gpurun_out/${tag}hbm_tcc -o p -- python $root/bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY --output-format csv -d $out/pmc_sq -o p -- python $root/bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline > /dev/null 2>&1
cd $root
python tools/hbm_traffic.py ${tag}hbm $out/hbm_traffic.json "round 2 kernels"
cp $out/hbm_traffic.json $root/profiles/${tag}_hbm_traffic.json
cd /tmp
python $root/bench.py --steps 20 --warmup 5 > $out/bench_vga.json 2> $out/bench_vga.err
python $root/bench.py --steps 20 --warmup 5 --workload 1280x960 --batch 256 --cpu-seconds 5 > $out/bench_1280x960.json 2> $out/bench_1280x960.err
python $root/bench.py --steps 20 --warmup 5 --workload 720p-build --batch 64 --cpu-seconds 5 > $out/bench_720p_build.json 2> $out/bench_720p_build.err
python $root/bench.py --steps 20 --warmup 5 --streams 1 --no-cpu-baseline > $out/bench_vga_streams1.json 2> /dev/null
python $root/bench.py --steps 20 --warmup 5 --log-bucket-size 4 --bucket-limit 3 --no-cpu-baseline > $out/bench_vga_buckets43.json 2> /dev/null
for w in vga 1280x960 720p-build; do
  b=256; [ $w = 720p-build ] && b=64
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$w -o p -- python $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $w --batch $b > $out/bench_under_rocprof_$w.json 2>/dev/null
  cp $out/trace_$w/p_kernel_stats.csv $out/kernel_stats_$w.csv
  # one batch at a time: per-kernel durations without another batch's kernels sharing the GPU (these are the
  # durations roofline.launch_ms has to agree with)
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace1_$w -o p -- python $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --streams 1 --workload $w --batch $b > /dev/null 2>&1
  cp $out/trace1_$w/p_kernel_stats.csv $out/kernel_stats_${w}_streams1.csv
done
cd $root
python tools/pmc_aggregate.py gpurun_out/${tag}hbm_fetch/p_counter_collection.csv $out/pmc_fetch_size.csv
python tools/pmc_aggregate.py gpurun_out/${tag}hbm_write/p_counter_collection.csv $out/pmc_write_size.csv
python tools/pmc_aggregate.py $out/pmc_sq/p_counter_collection.csv $out/pmc_sq_counters.csv
cat $out/pmc_sq_counters.csv | cut -d, -f1-4
rm -rf $out/trace_* $out/trace1_* $out/pmc_sq
ls -la $out
