mkdir -p gpurun_out/r2g
(time python -m pytest tests -m gpu -x -q) > gpurun_out/r2g/pytest.log 2>&1
grep -n "passed\|failed\|Error" gpurun_out/r2g/pytest.log | tail -5
python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r2g/vga.json 2>/dev/null
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2g/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], round(d['ms_per_step'],4), '%.3e'%d['value'], round(d['roofline']['launch_ms'],4), round(d['roofline']['frac'],4), {k:round(v,4) for k,v in d['roofline']['stage_ms'].items()})
P
