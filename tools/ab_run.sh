#!/bin/bash
# usage (GPU box): bash tools/ab_run.sh <variant> [<variant> ...] — development A/B of the variant libraries built by
# tools/ab_build.sh (variants/libpislam_hip_<variant>.so): three interleaved rounds of the default pipelined bench per variant
# (ms/step, kp+desc/s, strip-kernel ms, stage times) on the synthetic input AND on the reference's demo photo (no kernel change is
# accepted on the synthetic input alone), then the SQ instruction counters of the pf:: kernels per variant.
# Differences below ~0.5 % are box noise; run the GPU suite separately (PISLAM_HIP_LIB=... python -m pytest tests -m gpu).
run() { echo "$1: $(timeout 300 bash tools/bench_quick.sh ${@:2})"; }
for rep in 1 2 3; do
for n in "$@"; do
  export PISLAM_HIP_LIB=$PWD/variants/libpislam_hip_$n.so
  run "$n s3"
  run "$n s3 demo-photo" --workload demo-photo
done
done
for n in "$@"; do
  export PISLAM_HIP_LIB=$PWD/variants/libpislam_hip_$n.so
  echo "$n $(bash tools/pmc_quick.sh 2>&1 | grep 'k_fused_strips \|k_gather_orb ')"
done
