mkdir -p gpurun_out/r2s
for s in 1 2 3 4; do echo "vga S=$s"; bash tools/bench_quick.sh --streams $s; bash tools/bench_quick.sh --streams $s; done > gpurun_out/r2s/streams.log 2>&1
for s in 1 2 3; do echo "1280x960 S=$s"; bash tools/bench_quick.sh --streams $s --workload 1280x960; done >> gpurun_out/r2s/streams.log 2>&1
for s in 1 2 3; do echo "720p S=$s"; bash tools/bench_quick.sh --streams $s --workload 720p-build --batch 64; done >> gpurun_out/r2s/streams.log 2>&1
for s in 1 2; do echo "force S=$s"; bash tools/bench_quick.sh --streams $s --force-exchange; done >> gpurun_out/r2s/streams.log 2>&1
echo "2 ranks"; python bench.py --gpus 2 --dist-backend gloo --steps 20 --warmup 3 2>&1 | tail -2 >> gpurun_out/r2s/streams.log
python bench.py --steps 50 --warmup 5 --streams 2 --no-cpu-baseline > gpurun_out/r2s/err.log 2>&1
cat gpurun_out/r2s/streams.log
