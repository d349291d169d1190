mkdir -p gpurun_out/r5a
(timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "first_bucket_mode or hipgraph_capturable or demo_pyramid" 2>&1 | tail -5) > gpurun_out/r5a/pytest_new.log 2>&1
timeout 600 python bench.py --steps 50 --warmup 10 --cpu-seconds 5 > gpurun_out/r5a/bench_vga.json 2> gpurun_out/r5a/bench_vga.err; echo "rc=$?" >> gpurun_out/r5a/bench_vga.err
timeout 600 python bench.py --steps 50 --warmup 10 --workload demo-photo --cpu-seconds 3 > gpurun_out/r5a/bench_demo.json 2> gpurun_out/r5a/bench_demo.err; echo "rc=$?" >> gpurun_out/r5a/bench_demo.err
timeout 600 python bench.py --steps 30 --warmup 10 --workload 1280x960 --cpu-seconds 3 > gpurun_out/r5a/bench_1280.json 2> gpurun_out/r5a/bench_1280.err; echo "rc=$?" >> gpurun_out/r5a/bench_1280.err
timeout 600 python bench.py --steps 50 --warmup 10 --log-bucket-size 4 --bucket-limit 3 --no-cpu-baseline > gpurun_out/r5a/bench_b43.json 2> gpurun_out/r5a/bench_b43.err; echo "rc=$?" >> gpurun_out/r5a/bench_b43.err
timeout 600 python bench.py --steps 30 --warmup 10 --workload 720p-build --batch 64 --no-cpu-baseline > gpurun_out/r5a/bench_720.json 2> gpurun_out/r5a/bench_720.err; echo "rc=$?" >> gpurun_out/r5a/bench_720.err
(timeout 900 bash tools/pmc_ablate.sh --streams 1) > gpurun_out/r5a/phase_ablation.txt 2>&1
(timeout 600 bash tools/pmc_ablate.sh --streams 1 --workload demo-photo) > gpurun_out/r5a/phase_ablation_demo.txt 2>&1
rm -rf gpurun_out/abl_*
tail -3 gpurun_out/r5a/pytest_new.log; cat gpurun_out/r5a/*.err | tail -20; cat gpurun_out/r5a/phase_ablation.txt
