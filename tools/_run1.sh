mkdir -p gpurun_out/r2a
(time python -m pytest tests -m gpu -x -q) > gpurun_out/r2a/pytest.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r2a/bench_vga.json 2> gpurun_out/r2a/bench_vga.err
python bench.py --steps 20 --warmup 5 --workload 1280x960 --batch 64 --cpu-seconds 4 > gpurun_out/r2a/bench_1280.json 2> gpurun_out/r2a/bench_1280.err
python bench.py --steps 20 --warmup 5 --workload 720p-build --batch 64 --cpu-seconds 4 > gpurun_out/r2a/bench_720.json 2> gpurun_out/r2a/bench_720.err
tail -5 gpurun_out/r2a/pytest.log; cat gpurun_out/r2a/bench_vga.json | cut -c1-600; tail -3 gpurun_out/r2a/*.err
