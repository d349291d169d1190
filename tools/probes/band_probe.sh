#!/bin/bash
# rolling-band probe on the bench's synthetic pyramids, next to the strip kernel restricted to the same phases
# (profiling build, ablate 8 = staging + prefilter + pretest + FAST-9 + Harris, no NMS / emit)
python -c "
from pislam_amd import synth
synth.make_batch(0, 16).tofile('/tmp/pyr16.raw')"
for seg in 32 56 112; do timeout 120 tools/probes/_bin/band_probe /tmp/pyr16.raw 256 $seg | tail -1; done
bash tools/bench_quick.sh --ablate 8 --streams 1
bash tools/pmc_quick.sh --ablate 8 | grep strips
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $root/gpurun_out/bp -o p -- $root/tools/probes/_bin/band_probe /tmp/pyr16.raw 256 56 > /dev/null 2>&1
python - <<P
import csv, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open("$root/gpurun_out/bp/p_counter_collection.csv")):
    if "k_bands" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("k_bands", {c: round(sorted(v)[len(v)//2] / 1e6, 2) for c, v in agg.items()})
P
rm -rf $root/gpurun_out/bp
