mkdir -p gpurun_out/c4
(time timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5) > gpurun_out/c4/pytest.log 2>&1; tail -3 gpurun_out/c4/pytest.log
for rep in 1 2 3; do
for n in base grev; do
  export PISLAM_HIP_LIB=$PWD/variants/libpislam_hip_$n.so
  echo "$n s3: $(timeout 300 bash tools/bench_quick.sh)" | tee -a gpurun_out/c4/ab.txt
  echo "$n s1: $(timeout 300 bash tools/bench_quick.sh --streams 1)" | tee -a gpurun_out/c4/ab.txt
done
done
export PISLAM_HIP_LIB=$PWD/variants/libpislam_hip_base.so
for ch in 8 10 12 14 16; do
  echo "base chunks $ch: $(timeout 300 bash tools/bench_quick.sh --opt orb_chunks=$ch)" | tee -a gpurun_out/c4/ab.txt
done
