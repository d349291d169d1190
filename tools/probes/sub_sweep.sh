#!/bin/bash
# usage (GPU box, repo root): bash tools/probes/sub_sweep.sh <outdir>
# Sub-batch pipelining inside pislam_orb_frontend_batch (option sub_batches) x caller-side pipelines (--streams).
out=${1:-gpurun_out/sub_sweep}
mkdir -p $out
run() {  # label, args...
  local label=$1; shift
  python bench.py --steps 40 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label', 'ms/step %.4f' % d['ms_per_step'], 'value %.3e' % d['value'], 'strip_ms %.4f' % d['roofline']['launch_ms'], 'gpu_ms %.4f' % d['roofline']['step_gpu_ms'])
" | tee -a $out/sweep.txt
}
for st in 1 2 3; do for sb in 1 2 3 4 6 8; do run "vga256 streams=$st sub=$sb" --streams $st --sub-batches $sb; done; done
for sb in 1 4 8 12 16; do run "1280x960 b256 streams=1 sub=$sb" --workload 1280x960 --streams 1 --sub-batches $sb; done
run "1280x960 b256 streams=3 sub=1" --workload 1280x960 --streams 3 --sub-batches 1
run "1280x960 b256 streams=2 sub=8" --workload 1280x960 --streams 2 --sub-batches 8
for sb in 1 2 4; do run "720p b64 streams=1 sub=$sb" --workload 720p-build --batch 64 --streams 1 --sub-batches $sb; done
run "720p b64 streams=3 sub=1" --workload 720p-build --batch 64 --streams 3 --sub-batches 1
