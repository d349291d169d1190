#!/bin/bash
# usage (GPU box, repo root): bash tools/quick_gpu.sh <outdir> [notest] -- one development iteration:
# GPU suite, bench (one call at a time / default pipelines, one-grid step on / off), SQ counters of the pf:: kernels.
out=${1:-gpurun_out/quick}
mkdir -p $out
export PISLAM_FUSED_TRACE=1
if [ "$2" != "notest" ]; then
  (time timeout 900 python -m pytest tests -m gpu -x -q --durations=5) > $out/pytest.log 2>&1
  tail -12 $out/pytest.log
fi
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --streams 1 2>&1 | grep "fused grid"
for a in "--streams 1" "--streams 1 --opt fused_grid=0" "" "--opt fused_grid=0" "--streams 2" "--streams 1 --opt orb_lag=3" "--streams 1 --opt orb_lag=4"  "--streams 1 --opt orb_lag=8" "--streams 1 --opt orb_lag=12" "--streams 1 --opt orb_group=16 --opt orb_lag=4" "--streams 1 --opt orb_group=32 --opt orb_lag=2"; do
  echo "bench $a: $(timeout 300 bash tools/bench_quick.sh $a)" | tee -a $out/bench.txt
done
timeout 300 bash tools/pmc_quick.sh 2>&1 | tee -a $out/pmc.txt
