mkdir -p gpurun_out/r5f
(time timeout 1500 python -m pytest tests -m gpu -x -q --durations=3) > gpurun_out/r5f/pytest.log 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" gpurun_out/r5f/pytest.log | tail -8
(timeout 900 python tests/fuzz_campaign.py --seeds 4000 --start 3400000 | tail -1) 2>&1 | tee gpurun_out/r5f/fuzz.txt
python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r5f/bench_vga.json 2> gpurun_out/r5f/bench_vga.err
python - <<P
import json
d=json.loads(open("gpurun_out/r5f/bench_vga.json").read().strip().splitlines()[-1])
print("default: ms/step", d["ms_per_step"], "one_batch", d["one_batch_ms"], "one_pyr", d["one_pyramid_ms"], "latency", d["one_pyramid_latency_ms"], d["one_pyramid_path"], "parity", d["parity_in_run"]["ok"])
P
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --opt frame=0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('frame=0: one_pyr', d['one_pyramid_ms'], 'latency', d['one_pyramid_latency_ms'], d['one_pyramid_path'])"
cd /tmp && export TMPDIR=/tmp
for b in 1 2; do for f in 1 0; do
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5f/tr -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-one-pyramid --parity-pyramids 0 --batch $b --streams 1 --opt frame=$f > /dev/null 2>&1
python - <<P | tee -a $GRAFT_REPO_ROOT/gpurun_out/r5f/trace.txt
import csv
for r in list(csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/r5f/tr/p_kernel_stats.csv")))[:3]:
    if int(r["Calls"]) > 100: print("batch=$b frame=$f", r["Name"].split("(")[0][-40:], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
P
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r5f/tr
done; done
cd $GRAFT_REPO_ROOT
bash tools/asan_round.sh gpu gpurun_out/r5f/asan_gpu.txt > /dev/null 2>&1
tail -25 gpurun_out/r5f/asan_gpu.txt
