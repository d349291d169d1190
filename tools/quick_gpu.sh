#!/bin/bash
# usage (GPU box, repo root): bash tools/quick_gpu.sh <outdir> [notest] -- one development iteration:
# GPU suite, bench (one call at a time / batches in flight), SQ counters of the pf:: kernels.
out=${1:-gpurun_out/quick}
mkdir -p $out
if [ "$2" != "notest" ]; then
  (time timeout 900 python -m pytest tests -m gpu -x -q --durations=5) > $out/pytest.log 2>&1
  tail -12 $out/pytest.log
fi
for a in "--streams 1" "" "--streams 2"; do
  echo "bench $a: $(timeout 300 bash tools/bench_quick.sh $a)" | tee -a $out/bench.txt
done
timeout 300 bash tools/pmc_quick.sh 2>&1 | tee -a $out/pmc.txt
