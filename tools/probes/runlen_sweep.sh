#!/bin/bash
# x-tiles on VGA levels (taller strips at the cost of 32 halo columns per tile)
for tc in 256 320 448 512 640; do echo "vga tile_cols=$tc"; bash tools/bench_quick.sh --tile-cols $tc;  done
for tc in 320 512; do echo "vga tile_cols=$tc rows_max 64"; bash tools/bench_quick.sh --tile-cols $tc --strip-rows-max 64;  done
for tc in 320 448 512; do echo "1280 tile_cols=$tc"; bash tools/bench_quick.sh --tile-cols $tc --workload 1280x960;  done
