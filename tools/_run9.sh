python -m pytest tests/test_prep.py tests/test_gpu_fuzz.py -m gpu -x -q -k "pyramid or build or prep or gaussian or bilinear" 2>&1 | tail -3
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload 720p-build --batch 64 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['stage_ms'], d['roofline']['step_gpu_ms'])"
