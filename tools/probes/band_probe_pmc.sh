cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
python -c "
import sys; sys.path.insert(0, '$root')
from pislam_amd import synth
synth.make_batch(0, 16).tofile('/tmp/pyr16.raw')"
for seg in 112 224 448; do
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $root/gpurun_out/bp -o p -- $root/tools/probes/_bin/band_probe /tmp/pyr16.raw 256 $seg > /dev/null 2>&1
python - <<P
import csv, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open("$root/gpurun_out/bp/p_counter_collection.csv")):
    if "k_bands" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("seg $seg k_bands", {c: round(sorted(v)[len(v)//2] / 1e6, 2) for c, v in agg.items()})
P
rm -rf $root/gpurun_out/bp
done
