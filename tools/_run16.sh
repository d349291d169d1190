q() { echo -n "$* : "; bash tools/_q.sh "$@" | cut -c1-60; }
q; q --batch 128; q --batch 64; q --batch 32; q --workload 1280x960 --batch 256; q --workload 1280x960 --batch 64; q --workload 720p-build --batch 64; q --workload 720p-build --batch 256; q --log-bucket-size 4 --bucket-limit 3
