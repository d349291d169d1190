#!/bin/bash
# strip height of the strip kernel re-tuned (rows_max = cap of the height heuristic on the narrow levels)
for rm in 32 36 40 44 48 56 64; do echo "vga rows_max=$rm"; bash tools/bench_quick.sh --strip-rows-max $rm;  bash tools/bench_quick.sh --strip-rows-max $rm --streams 1; done
for rm in 36 44; do for rl in 1 3 4; do echo "vga rows_max=$rm run_len=$rl"; bash tools/bench_quick.sh --strip-rows-max $rm --run-len $rl; done; done
for rm in 36 44 56; do echo "1280x960 rows_max=$rm"; bash tools/bench_quick.sh --strip-rows-max $rm --workload 1280x960; done
for rm in 36 44 56; do echo "720p rows_max=$rm"; bash tools/bench_quick.sh --strip-rows-max $rm --workload 720p-build --batch 64; done
