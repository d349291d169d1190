#!/bin/bash
for rep in 1 2; do
for n in 0 7 8 10 12 14 16 20 28; do
  echo "orb_chunks $n: $(timeout 300 bash tools/bench_quick.sh --opt orb_chunks=$n)"
done
done
