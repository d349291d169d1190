#!/bin/bash
# development A/B of the variant libraries under variants/ (see tools/ab_build.sh); the working tree's library runs the GPU suite
true
run() { echo "$1: $(timeout 300 bash tools/bench_quick.sh ${@:2})"; }
for rep in 1 2 3; do
for n in "$@"; do
  export PISLAM_HIP_LIB=$PWD/variants/libpislam_hip_$n.so
  run "$n s3"
done
done
for n in "$@"; do
  export PISLAM_HIP_LIB=$PWD/variants/libpislam_hip_$n.so
  echo "$n $(bash tools/pmc_quick.sh 2>&1 | grep 'k_fused_strips \|k_gather_orb ')"
done
