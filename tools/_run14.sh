for rl in 4 6 8 10 12 16; do echo -n "run_len $rl: "; bash tools/_q.sh --run-len $rl; done
for sr in 16 20 22 24 26 28 32; do echo -n "strip_rows $sr: "; bash tools/_q.sh --strip-rows $sr; done
