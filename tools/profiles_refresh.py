#!/usr/bin/env python3
"""usage (dev container, after `gpurun -- bash tools/profile_round.sh r02`): python tools/profiles_refresh.py [r02]
Copies the round's results from gpurun_out/<tag>/ into profiles/<tag>_* under the names profiles/README.md lists and
prints the numbers the docs quote (ms/step, kp+desc/s, strip-kernel ms, roofline fraction, kernel-trace averages)."""
import csv
import json
import os
import shutil
import subprocess
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")
names = {
    "bench_vga.json": "bench_vga.json", "bench_vga_streams1.json": "bench_vga_streams1.json",
    "bench_1280x960.json": "bench_1280x960_batch256.json", "bench_720p_build.json": "bench_720p_build_batch64.json",
    "bench_vga_buckets43.json": "bench_vga_buckets43.json", "bench_under_rocprof_vga.json": "bench_under_rocprof_vga.json",
    "kernel_stats_vga.csv": "kernel_stats_vga.csv", "kernel_stats_vga_streams1.csv": "kernel_stats_vga_streams1.csv",
    "kernel_stats_1280x960.csv": "kernel_stats_1280x960_batch256.csv",
    "kernel_stats_1280x960_streams1.csv": "kernel_stats_1280x960_batch256_streams1.csv",
    "kernel_stats_720p-build.csv": "kernel_stats_720p_build_batch64.csv",
    "kernel_stats_720p-build_streams1.csv": "kernel_stats_720p_build_batch64_streams1.csv",
    "hbm_traffic.json": "hbm_traffic.json", "pmc_fetch_size.csv": "pmc_fetch_size.csv",
    "pmc_write_size.csv": "pmc_write_size.csv", "pmc_sq_counters.csv": "pmc_sq_counters.csv",
}
for a, b in names.items():
    shutil.copyfile(os.path.join(src, a), os.path.join(dst, f"{tag}_{b}"))
with open(os.path.join(dst, f"{tag}_kernel_resources.txt"), "w") as f:
    f.write(subprocess.run(["bash", os.path.join(root, "tools", "kernel_resources.sh")], capture_output=True, text=True).stdout)
for b in ("bench_vga", "bench_vga_streams1", "bench_1280x960_batch256", "bench_720p_build_batch64", "bench_vga_buckets43"):
    d = json.loads(open(os.path.join(dst, f"{tag}_{b}.json")).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f"{b}: {d['ms_per_step']:.4f} ms/step  {d['value']:.4e} kp+desc/s  strips {r['launch_ms']:.4f} ms  frac {r['frac']:.4f}  "
          f"stages { {k: round(v, 4) for k, v in r['stage_ms'].items()} }  cpu {d.get('cpu_baseline', {}).get('value')}"
          f" / {d.get('cpu_baseline', {}).get('all_threads', {}).get('value') if isinstance(d.get('cpu_baseline', {}).get('all_threads'), dict) else ''}")
for b in ("kernel_stats_vga_streams1", "kernel_stats_1280x960_batch256_streams1", "kernel_stats_720p_build_batch64_streams1"):
    rows = list(csv.DictReader(open(os.path.join(dst, f"{tag}_{b}.csv"))))[:7]
    print(b, [(r["Name"].split("(")[0].split("::")[-1][:24], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1)) for r in rows])
tj = json.load(open(os.path.join(dst, f"{tag}_hbm_traffic.json")))
print("hbm", tj.get("source_hash"), {k: round(v.get("hbm_bytes_per_launch", 0) / 1e6, 1) for k, v in tj["kernels"].items()})
for r in csv.DictReader(open(os.path.join(dst, f"{tag}_pmc_sq_counters.csv"))):
    if r["Counter_Name"] in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
        print(r["Kernel_Name"][:40], r["Counter_Name"], round(float(r["mean"]) / 1e6, 2))
sys.path.insert(0, root)
from pislam_amd import build  # noqa: E402
print("source_hash now", build.source_hash())
