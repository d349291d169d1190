#!/bin/bash
out=${1:-gpurun_out/expsel}; mkdir -p $out
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^$" | tail -25) | tee $out/pytest.txt
run() { echo "$1: $(timeout 300 bash tools/bench_quick.sh ${@:2})" | tee -a $out/res.txt; }
B="--log-bucket-size 4 --bucket-limit 3"
for rep in 1 2; do
run "select s1" $B --streams 1
run "instrip s1" $B --streams 1 --opt bucket_select=0
run "select s3" $B
run "instrip s3" $B --opt bucket_select=0
done
run "plain s3"
