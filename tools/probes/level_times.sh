# per-dispatch durations of the pyramid build kernels of one 720p-build step (GPU box, repo root): DESIGN.md section 8
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr720 -o p -- python $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --streams 1 --graph 0 --workload 720p-build --batch 64 --spin-s 0.3 > /dev/null 2>&1
python - <<P
import csv
rows = sorted(csv.DictReader(open("/tmp/tr720/p_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
for r in [r for r in rows if "pp::" in r["Kernel_Name"]][-18:]:
    print(r["Kernel_Name"].split("(")[0].split("::")[-1][:28], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000, "us")
P
