mkdir -p gpurun_out/r2b
(time python -m pytest tests -m gpu -x -q) > gpurun_out/r2b/pytest.log 2>&1
tail -5 gpurun_out/r2b/pytest.log
for o in 1 0; do python bench.py --steps 50 --warmup 5 --no-cpu-baseline --orb-in-strip $o > gpurun_out/r2b/bench_orb$o.json 2> gpurun_out/r2b/bench_orb$o.err; done
python - <<'P'
import json
for o in (1,0):
    d=json.loads(open(f'gpurun_out/r2b/bench_orb{o}.json').read().strip().splitlines()[-1])
    print(o, d['ms_per_step'], d['value'], d['roofline']['launch_ms'], d['roofline']['stage_ms'])
P
