mkdir -p gpurun_out/r5d
(time timeout 1500 python -m pytest tests -m gpu -x -q --durations=3) > gpurun_out/r5d/pytest.log 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" gpurun_out/r5d/pytest.log | tail -25
(timeout 900 python tests/fuzz_campaign.py --seeds 3000 --wide --start 3200000 | tail -2) 2>&1 | tee gpurun_out/r5d/fuzz.txt
(timeout 900 python tests/fuzz_campaign.py --seeds 6000 --start 3300000 | tail -2) 2>&1 | tee -a gpurun_out/r5d/fuzz.txt
q() { python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-one-pyramid --parity-pyramids 4 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), 'one_batch', round(d['one_batch_ms'],4), d.get('parity_in_run',{}).get('ok'))"; }
for b in 1 2 4 8 16; do
  echo "batch $b frame1: $(q --batch $b)" | tee -a gpurun_out/r5d/small.txt
  echo "batch $b frame0: $(q --batch $b --opt frame=0)" | tee -a gpurun_out/r5d/small.txt
done
echo "batch 1 frame1 demo: $(q --batch 1 --workload demo-photo)" | tee -a gpurun_out/r5d/small.txt
echo "batch 1 frame0 demo: $(q --batch 1 --workload demo-photo --opt frame=0)" | tee -a gpurun_out/r5d/small.txt
echo "default: $(timeout 300 bash tools/bench_quick.sh)" | tee -a gpurun_out/r5d/small.txt
