mkdir -p gpurun_out/r2f
(time python -m pytest tests -m gpu -x -q) > gpurun_out/r2f/pytest.log 2>&1
grep -n "passed\|failed\|Error\|assert" gpurun_out/r2f/pytest.log | tail -15
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r2f/vga.json 2>/dev/null
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload 1280x960 --batch 64 > gpurun_out/r2f/w1280.json 2>/dev/null
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload 1280x960 --batch 256 > gpurun_out/r2f/w1280_b256.json 2>/dev/null
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload 720p-build --batch 64 > gpurun_out/r2f/b720.json 2>/dev/null
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2f/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['ms_per_step'],4), '%.3e'%d['value'], round(d['roofline']['launch_ms'],4), round(d['roofline']['frac'],4), {k:round(v,4) for k,v in d['roofline']['stage_ms'].items()})
    except Exception as e: print(f, 'ERR', e)
P
