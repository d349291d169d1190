bash tools/_q.sh; bash tools/_q.sh
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | grep -n "passed\|failed" | tail -2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcq -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python - <<P
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/pmcq/p_counter_collection.csv")):
    k = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
    if k in ("k_fused_strips","k_gather_orb"): agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sorted(v)[len(v)//2] / 1e6, 2) for c, v in d.items()})
P
