mkdir -p gpurun_out/c8
(time timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/c8/pytest.log 2>&1; grep -n "passed\|failed" gpurun_out/c8/pytest.log | tail -3
for rep in 1 2 3; do
for n in base merge; do
  export PISLAM_HIP_LIB=$PWD/variants/libpislam_hip_$n.so
  echo "$n s3: $(timeout 300 bash tools/bench_quick.sh)" | tee -a gpurun_out/c8/ab.txt
  echo "$n demo: $(timeout 300 bash tools/bench_quick.sh --workload demo-photo)" | tee -a gpurun_out/c8/ab.txt
done
done
for n in base merge; do
  export PISLAM_HIP_LIB=$PWD/variants/libpislam_hip_$n.so
  echo "$n 1280: $(timeout 300 bash tools/bench_quick.sh --workload 1280x960)" | tee -a gpurun_out/c8/ab.txt
  echo "$n 720p: $(timeout 300 bash tools/bench_quick.sh --workload 720p-build --batch 64)" | tee -a gpurun_out/c8/ab.txt
  echo "$n $(bash tools/pmc_quick.sh 2>&1 | grep 'k_fused_strips ')" | tee -a gpurun_out/c8/ab.txt
done
unset PISLAM_HIP_LIB
timeout 600 python tests/fuzz_campaign.py --seeds 6000 --wide --start 3900000 | tail -1 | tee -a gpurun_out/c8/ab.txt
