mkdir -p gpurun_out/r2c
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --orb-in-strip 0 --graph 0 --ablate 8192 > gpurun_out/r2c/prof0.json 2> gpurun_out/r2c/prof0.err
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --orb-in-strip 1 --graph 0 --ablate 8192 > gpurun_out/r2c/prof1.json 2> gpurun_out/r2c/prof1.err
grep "pislam prof" gpurun_out/r2c/prof0.err | tail -2; grep "pislam prof" gpurun_out/r2c/prof1.err | tail -2
bash tools/pmc_hbm.sh r2c_hbm1 --orb-in-strip 1
python tools/hbm_traffic.py r2c_hbm1 gpurun_out/r2c/hbm_orb1.json "in-strip ORB"
