mkdir -p gpurun_out/r5b
(time timeout 1500 python -m pytest tests -m gpu -x -q --durations=5) > gpurun_out/r5b/pytest.log 2>&1
tail -12 gpurun_out/r5b/pytest.log
bash tools/ab_run.sh base ovl noovl pkmin pkw 2>&1 | tee gpurun_out/r5b/ab.txt
for n in base ovl; do
  export PISLAM_HIP_LIB=$PWD/variants/libpislam_hip_$n.so
  echo "$n demo: $(timeout 300 bash tools/bench_quick.sh --workload demo-photo)" | tee -a gpurun_out/r5b/ab.txt
  echo "$n rl3: $(timeout 300 bash tools/bench_quick.sh --run-len 3)" | tee -a gpurun_out/r5b/ab.txt
  echo "$n rl4: $(timeout 300 bash tools/bench_quick.sh --run-len 4)" | tee -a gpurun_out/r5b/ab.txt
  echo "$n cycles: $(python bench.py --steps 3 --warmup 1 --streams 1 --graph 0 --no-cpu-baseline --no-one-pyramid --parity-pyramids 0 --ablate 8192 2>&1 | grep 'cycles/strip' | head -1)" | tee -a gpurun_out/r5b/ab.txt
done
