mkdir -p gpurun_out/r5e
export PISLAM_HIP_LIB=$PWD/variants/libpislam_hip_fr16.so
cd /tmp && export TMPDIR=/tmp
for f in 1 0; do
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5e/tr$f -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-one-pyramid --parity-pyramids 0 --batch 1 --streams 1 --opt frame=$f > /dev/null 2>&1
python - <<P | tee -a $GRAFT_REPO_ROOT/gpurun_out/r5e/trace.txt
import csv
for r in list(csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/r5e/tr$f/p_kernel_stats.csv")))[:5]:
    print("frame=$f", r["Name"].split("(")[0][-40:], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
P
done
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r5e/tr*
