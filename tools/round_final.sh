#!/bin/bash
# usage (GPU box): bash tools/round_final.sh <tag> — the round's closing run: GPU suite, fuzz campaign on the final sources
# (hash recorded), then tools/profile_round.sh
tag=${1:-r06}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag; mkdir -p $out
cd $root
for i in 1 2 3; do (time timeout 1500 python -m pytest tests -m gpu -q --durations=5) > $out/pytest_gpu_$i.log 2>&1; grep "passed\|failed" $out/pytest_gpu_$i.log | tail -1; done
cp $out/pytest_gpu_1.log $out/pytest_gpu.log
h=$(python -c "from pislam_amd import build; print(build.source_hash())")
{
  echo "# python tests/fuzz_campaign.py --seeds 20000 --wide --start 3600000; --seeds 60000 --start 3700000 (kernel sources $h)"
  timeout 1200 python tests/fuzz_campaign.py --seeds 20000 --wide --start 3600000 | tail -3
  timeout 1200 python tests/fuzz_campaign.py --seeds 60000 --start 3700000 | tail -3
} > $out/fuzz.txt 2>&1
tail -2 $out/fuzz.txt
# the host code under UBSan traps against the GPU suites (build variants/ first, here: tools/asan_round.sh build)
[ -f variants/libpislam_hip_ubsan.so ] && bash tools/asan_round.sh gpu $out/ubsan_gpu.txt > /dev/null 2>&1
bash tools/profile_round.sh $tag > $out/profile_round.log 2>&1
tail -5 $out/profile_round.log
