#!/bin/bash
# matcher A/B: matrix-core kernel vs VALU popcount kernel inside the bench's --match step
for m in 1 0; do python bench.py --steps 10 --warmup 2 --no-cpu-baseline --match --match-mfma $m 2>/dev/null | grep "^{" | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('mfma=$m', d['ms_per_step'], d['config']['match_inside_step'])"; done
