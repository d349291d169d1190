#!/bin/bash
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep "passed\|failed\|Error" | tail -3)
for rep in 1 2; do
for n in 0 22 20 18; do
  echo "strip_rows $n: $(timeout 300 bash tools/bench_quick.sh --opt strip_rows=$n)"
done
done
