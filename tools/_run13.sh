mkdir -p gpurun_out/r2h
python -m pytest tests/test_prep.py tests/test_gpu_fuzz.py -m gpu -x -q -k "pyramid or build or prep or gaussian or bilinear or single_level" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2h/prof -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload 720p-build --batch 64 > $GRAFT_REPO_ROOT/gpurun_out/r2h/b720.json 2>/dev/null
python - <<P
import csv
for r in csv.DictReader(open('$GRAFT_REPO_ROOT/gpurun_out/r2h/prof/p_kernel_stats.csv')):
    if 'pp::' in r['Name']: print(r['Name'][:60], r['Calls'], r['AverageNs'])
P
