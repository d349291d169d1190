#!/bin/bash
# development: A/B of library variants (variants/libpislam_hip_<name>.so, tools/ab_build.sh) — usage: tools/exp_ab.sh <out> <name>... [-- bench args]
out=$1; shift; mkdir -p $out
names=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do names+=("$1"); shift; done; [ "$1" = "--" ] && shift
run() { echo "$1: $(timeout 300 bash tools/bench_quick.sh ${@:2})" | tee -a $out/res.txt; }
for rep in 1 2; do
for n in base "${names[@]}"; do
  if [ $n = base ]; then unset PISLAM_HIP_LIB; else export PISLAM_HIP_LIB=variants/libpislam_hip_$n.so; fi
  run "$n s1" --streams 1 "$@"
  run "$n s3" "$@"
done
done
