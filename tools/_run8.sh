mkdir -p gpurun_out/r2h
python -m pytest tests/test_prep.py tests/test_gpu_fuzz.py -m gpu -x -q -k "pyramid or build or prep or gaussian or bilinear" 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
for bp in 1 0; do
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2h/prof_bp$bp -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload 720p-build --batch ${BATCH:-64} --build-pairs $bp > $GRAFT_REPO_ROOT/gpurun_out/r2h/b720_bp$bp.json 2>/dev/null
python - <<P
import csv
rows=list(csv.DictReader(open('$GRAFT_REPO_ROOT/gpurun_out/r2h/prof_bp$bp/p_kernel_stats.csv')))
tot=0
for r in rows:
    n=r['Name']
    if 'pp::' in n:
        print($bp, n[:70], r['Calls'], r['AverageNs'])
        tot+=float(r['AverageNs'])*int(r['Calls'])
print('build total ns per step', tot/ (int([r for r in rows if 'k_gaussian' in r['Name']][0]['Calls'])))
P
done
