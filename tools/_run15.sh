for b in 64 32; do for rl in 1 2 3 4; do echo -n "batch $b run_len $rl: "; bash tools/_q.sh --batch $b --run-len $rl; done; done
for rl in 6 8 10 12; do echo -n "1280 b256 run_len $rl: "; bash tools/_q.sh --workload 1280x960 --batch 256 --run-len $rl; done
for rl in 4 6 8 10; do echo -n "720p b64 run_len $rl: "; bash tools/_q.sh --workload 720p-build --batch 64 --run-len $rl; done
