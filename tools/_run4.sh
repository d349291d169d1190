mkdir -p gpurun_out/r2d
for w in 0 3 4 5; do python bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload 1280x960 --batch 256 --wgs-per-cu $w > gpurun_out/r2d/w1280_b256_$w.json 2>/dev/null; done
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload 1280x960 --batch 256 --run-len 4 > gpurun_out/r2d/w1280_b256_rl4.json 2>/dev/null
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload 1280x960 --batch 64 --run-len 8 > gpurun_out/r2d/w1280_b64_rl8.json 2>/dev/null
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload 720p-build --batch 256 > gpurun_out/r2d/b720_b256.json 2>/dev/null
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2d/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['ms_per_step'],4), '%.3e'%d['value'], round(d['roofline']['launch_ms'],4), round(d['roofline']['frac'],4), {k:round(v,4) for k,v in d['roofline']['stage_ms'].items()})
    except Exception as e: print(f, 'ERR', e)
P
