q() { echo -n "$* : "; bash tools/_q.sh "$@" | cut -c1-60; }
q --workload 1280x960 --batch 256; q --workload 1280x960 --batch 64; q --workload 720p-build --batch 64; q
q --workload 1280x960 --batch 256 --tile-cols 600; q --workload 720p-build --batch 64 --tile-cols 600
