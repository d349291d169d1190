#!/bin/bash
# usage: tools/bench_quick.sh [bench.py args]  — one short bench run, prints ms/step, kp+desc/s, strip-kernel ms and stage times
# quick A/B: bench vga default (+ optional extra args), prints ms/step and stage times
python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-one-pyramid "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), '%.4e'%d['value'], round(d['roofline']['launch_ms'],4), {k:round(v,4) for k,v in d['roofline']['stage_ms'].items()})"
