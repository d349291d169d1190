mkdir -p gpurun_out/c7
(time timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/c7/pytest.log 2>&1; grep -n "passed\|failed" gpurun_out/c7/pytest.log | tail -3
for rep in 1 2 3; do
for n in base st7 st6 st8; do
  export PISLAM_HIP_LIB=$PWD/variants/libpislam_hip_$n.so
  echo "$n s3: $(timeout 300 bash tools/bench_quick.sh)" | tee -a gpurun_out/c7/ab.txt
done
done
for n in base st7 st6; do
  export PISLAM_HIP_LIB=$PWD/variants/libpislam_hip_$n.so
  echo "$n demo: $(timeout 300 bash tools/bench_quick.sh --workload demo-photo)" | tee -a gpurun_out/c7/ab.txt
  echo "$n 1280: $(timeout 300 bash tools/bench_quick.sh --workload 1280x960)" | tee -a gpurun_out/c7/ab.txt
done
export PISLAM_HIP_LIB=$PWD/variants/libpislam_hip_st7.so
timeout 600 bash tools/probes/tcp_counters.sh 2>&1 | grep gather | tee -a gpurun_out/c7/ab.txt
