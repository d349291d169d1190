#!/bin/bash
(timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep "passed\|failed")
run() { echo "$1: $(timeout 300 bash tools/bench_quick.sh ${@:2})"; }
for rep in 1 2 3; do
for n in prev new; do
  if [ $n = new ]; then unset PISLAM_HIP_LIB; else export PISLAM_HIP_LIB=variants/libpislam_hip_$n.so; fi
  run "$n s1" --streams 1
  run "$n s3"
done
done
unset PISLAM_HIP_LIB
bash tools/pmc_quick.sh 2>&1 | grep "k_fused_strips "
