mkdir -p gpurun_out/r5c
(time timeout 1500 python -m pytest tests -m gpu -x -q --durations=3) > gpurun_out/r5c/pytest.log 2>&1
tail -8 gpurun_out/r5c/pytest.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr"
(timeout 900 python tests/fuzz_campaign.py --seeds 6000 --wide --start 3100000 | tail -3) 2>&1 | tee gpurun_out/r5c/fuzz.txt
run() { echo "$1: $(timeout 300 bash tools/bench_quick.sh ${@:2})"; }
for rep in 1 2; do
for n in base ovl dfr nodfr; do
  export PISLAM_HIP_LIB=$PWD/variants/libpislam_hip_$n.so
  run "$n" | tee -a gpurun_out/r5c/ab.txt
done
done
for n in base dfr; do
  export PISLAM_HIP_LIB=$PWD/variants/libpislam_hip_$n.so
  for rl in 3 4 6; do run "$n rl$rl" --run-len $rl | tee -a gpurun_out/r5c/ab.txt; done
  run "$n demo" --workload demo-photo | tee -a gpurun_out/r5c/ab.txt
  run "$n demo rl4" --workload demo-photo --run-len 4 | tee -a gpurun_out/r5c/ab.txt
  run "$n 1280" --workload 1280x960 --steps 30 | tee -a gpurun_out/r5c/ab.txt
  run "$n 720" --workload 720p-build --batch 64 | tee -a gpurun_out/r5c/ab.txt
  run "$n 720 rl3" --workload 720p-build --batch 64 --run-len 3 | tee -a gpurun_out/r5c/ab.txt
  run "$n b43" --log-bucket-size 4 --bucket-limit 3 | tee -a gpurun_out/r5c/ab.txt
  echo "$n cycles: $(python bench.py --steps 3 --warmup 1 --streams 1 --graph 0 --no-cpu-baseline --no-one-pyramid --parity-pyramids 0 --ablate 8192 2>&1 | grep 'cycles/strip' | head -1)" | tee -a gpurun_out/r5c/ab.txt
  echo "$n $(bash tools/pmc_quick.sh 2>&1 | grep 'k_fused_strips ')" | tee -a gpurun_out/r5c/ab.txt
done
