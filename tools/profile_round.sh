#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>   e.g. r03
# Produces under gpurun_out/<tag>/ everything profiles/ keeps for a round — per workload the PMC counters
# (separate --pmc passes: FETCH_SIZE | WRITE_SIZE | TCC | SQ instructions | SQ LDS / activity), the bench lines (which quote those counters when the
# kernel sources match), rocprofv3 --kernel-trace --stats of the same commands — and copies the counter JSONs to
# profiles/ so that the bench lines of THIS run can already quote them.
tag=${1:-r06}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
pmc_passes() {   # <name> <workload> <batch> <bench args...>: four separate --pmc passes -> counters_<name>.json (+ profiles/)
  name=$1; w=$2; b=$3; shift 3
  args="--steps 3 --warmup 1 --graph 0 --no-cpu-baseline --no-one-pyramid --parity-pyramids 0 --spin-s 0.2 --workload $w --batch $b $*"
  i=0
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_INST_CYCLES_SALU"; do
    rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $out/pmc_${name}_$i -o p -- python $root/bench.py $args > /dev/null 2>&1
    i=$((i+1))
  done
  (cd $root && python tools/counters_json.py $out/counters_$name.json $w $b "bench.py $args" $out/pmc_${name}_0 $out/pmc_${name}_1 $out/pmc_${name}_2 $out/pmc_${name}_3 $out/pmc_${name}_4)
  cp $out/counters_$name.json $root/profiles/${tag}_counters_$name.json
  rm -rf $out/pmc_${name}_*
}
for w in vga 1280x960 720p-build demo-photo; do
  b=256; [ $w = 720p-build ] && b=64
  pmc_passes $w $w $b --streams 1
done
# the headline configuration itself: three batches in flight, every lane its own input (HBM bytes / L2 hits per step then)
pmc_passes vga_streams3 vga 256 --streams 3
# the README bucket mode <4,3>
pmc_passes vga_buckets43 vga 256 --streams 1 --log-bucket-size 4 --bucket-limit 3
cd /tmp
python $root/bench.py --steps 50 --warmup 10 > $out/bench_vga.json 2> $out/bench_vga.err
python $root/bench.py --steps 50 --warmup 10 --workload 1280x960 --batch 256 --cpu-seconds 5 > $out/bench_1280x960.json 2> $out/bench_1280x960.err
python $root/bench.py --steps 50 --warmup 10 --workload 720p-build --batch 64 --cpu-seconds 5 > $out/bench_720p-build.json 2> $out/bench_720p-build.err
python $root/bench.py --steps 50 --warmup 10 --workload demo-photo --cpu-seconds 5 > $out/bench_demo_photo.json 2> $out/bench_demo_photo.err
python $root/bench.py --steps 50 --warmup 10 --workload 1280x960-dense --batch 256 --no-cpu-baseline > $out/bench_1280x960-dense.json 2> /dev/null
python $root/bench.py --steps 50 --warmup 10 --streams 1 --no-cpu-baseline > $out/bench_vga_streams1.json 2> /dev/null
python $root/bench.py --steps 50 --warmup 10 --shared-input --no-cpu-baseline > $out/bench_vga_shared_input.json 2> /dev/null
python $root/bench.py --steps 50 --warmup 10 --no-cpu-baseline > $out/bench_vga_one_pyramid_per_call.json 2> /dev/null   # (one_pyramid_ms is part of the default line)
python $root/bench.py --steps 50 --warmup 10 --log-bucket-size 4 --bucket-limit 3 --no-cpu-baseline > $out/bench_vga_buckets43.json 2> /dev/null
python $root/bench.py --steps 50 --warmup 10 --log-bucket-size 4 --bucket-limit 3 --no-cpu-baseline --opt bucket_select=0 > $out/bench_vga_buckets43_in_strip_selection.json 2> /dev/null
python $root/bench.py --gpus 2 --dist-backend gloo --steps 20 --warmup 5 --batch 128 --no-cpu-baseline > $out/bench_vga_2ranks_one_gpu_gloo.json 2> $out/bench_2ranks.err
python $root/bench.py --gpus 8 --dist-backend gloo --steps 20 --warmup 5 --batch 32 --no-cpu-baseline > $out/bench_vga_8ranks_one_gpu_gloo.json 2> /dev/null
for w in vga 1280x960 720p-build demo-photo; do
  b=256; [ $w = 720p-build ] && b=64
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$w -o p -- python $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-one-pyramid --workload $w --batch $b > $out/bench_under_rocprof_$w.json 2>/dev/null
  cp $out/trace_$w/p_kernel_stats.csv $out/kernel_stats_$w.csv
  # one batch at a time: per-kernel durations without another batch's kernels sharing the GPU (these are the
  # durations roofline.launch_ms has to agree with)
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace1_$w -o p -- python $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-one-pyramid --streams 1 --workload $w --batch $b > /dev/null 2>&1
  cp $out/trace1_$w/p_kernel_stats.csv $out/kernel_stats_${w}_streams1.csv
done
python $root/tools/probes/pipeline_api.py 1 2 3 > $out/pipeline_api.txt 2>/dev/null
rm -rf $out/trace_* $out/trace1_*
ls -la $out
# the C++ host (no Python / torch in the process) on the reference's demo pyramid x 256: pislam_pipeline, 3 and 1 lanes
python - <<P
import numpy as np
z = np.load("$root/tests/golden/demo_pyramid.npz")
z["img"].astype(np.uint8).tofile("/tmp/demo_pyramid.raw")
P
(cd $root && make -s -C tools pislam_demo > /dev/null 2>&1; for s in 3 1; do tools/pislam_demo /tmp/demo_pyramid.raw --batch 256 --steps 100 --streams $s; done) > $out/cpp_tool_demo_photo_x256.txt 2>&1
# ... and a frame at a time (the reference's own use): one pyramid per call, one launch (pf::k_frame) vs three, 1 and 3 calls in flight
(cd $root && for s in 1 3; do for f in 1 0; do echo "streams $s frame=$f: $(tools/pislam_demo /tmp/demo_pyramid.raw --batch 1 --steps 3000 --streams $s --opt frame=$f 2>&1 | head -1)"; done; done) > $out/cpp_tool_one_pyramid_per_call.txt 2>&1
# kernel durations of one pyramid per call (one launch vs three)
for f in 1 0; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_one_$f -o p -- python $root/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-one-pyramid --parity-pyramids 0 --batch 1 --streams 1 --opt frame=$f > /dev/null 2>&1
  python - <<P >> $out/one_pyramid_kernel_trace.txt
import csv
for r in list(csv.DictReader(open("$out/trace_one_$f/p_kernel_stats.csv")))[:4]:
    if int(r["Calls"]) > 100: print("frame=$f", r["Name"].split("(")[0][-44:], "calls", r["Calls"], "avg ns", r["AverageNs"], "min", r["MinNs"], "max", r["MaxNs"])
P
  rm -rf $out/trace_one_$f
done
# per-phase instruction counts of the strip kernel (cumulative ablations of the profiling build)
(cd $root && bash tools/pmc_ablate.sh --streams 1) > $out/phase_ablation.txt 2>&1
(cd $root && bash tools/pmc_ablate.sh --streams 1 --log-bucket-size 4 --bucket-limit 3 --opt bucket_select=0) > $out/phase_ablation_buckets43_in_strip_selection.txt 2>&1
rm -rf $root/gpurun_out/abl_*
# per-phase counters of pf::k_gather_orb (cumulative ablations of its profiling instantiation)
(cd $root && bash tools/pmc_ablate_gather.sh) > $out/phase_ablation_gather.txt 2>&1
# round 6 probes: the one-launch frame path under other strip plans; the one-launch pyramid build with parts of its protocol off
(cd $root && bash tools/probes/frame_sweep.sh) > $out/frame_plan_sweep.txt 2>&1
(cd $root && python tools/probes/chain_probe.py 2>&1 | grep "us per build\|fault") > $out/build_chain_probe.txt
# workgroup wall-clock share of the strip kernel's phases (clock64 around the phases of every strip, profiling build):
# cycles per strip of one eager launch (the first lines: the later ones come from bench.py's 16-launch bracket)
python $root/bench.py --steps 3 --warmup 1 --streams 1 --graph 0 --no-cpu-baseline --ablate 8192 2>&1 | grep "cycles/strip" | head -3 > $out/phase_cycles.txt
python $root/bench.py --steps 3 --warmup 1 --streams 1 --graph 0 --no-cpu-baseline --ablate 8192 --log-bucket-size 4 --bucket-limit 3 --opt bucket_select=0 2>&1 | grep "cycles/strip" | head -3 > $out/phase_cycles_buckets43_in_strip_selection.txt
