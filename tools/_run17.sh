for rl in 1 2 3 5 7 9 10 11; do echo -n "vga b256 run_len $rl: "; bash tools/_q.sh --run-len $rl; done
for rl in 1 2 3 4 5; do echo -n "720p b64 run_len $rl: "; bash tools/_q.sh --workload 720p-build --batch 64 --run-len $rl; done
for rl in 3 4 5 6 7; do echo -n "1280 b256 run_len $rl: "; bash tools/_q.sh --workload 1280x960 --batch 256 --run-len $rl; done
