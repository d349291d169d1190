#!/bin/bash
for rep in 1 2; do
for w in 720p-build 1280x960; do
for n in 0 3 4 12; do
  echo "$w run_len $n: $(timeout 300 bash tools/bench_quick.sh --workload $w --opt run_len=$n)"
done
done
done
