#!/usr/bin/env python3
"""usage (dev container, after `gpurun -- bash tools/profile_round.sh r03`): python tools/profiles_refresh.py [r03]
Copies the round's results from gpurun_out/<tag>/ into profiles/<tag>_* and prints the numbers the docs quote."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")
for f in sorted(glob.glob(os.path.join(src, "*"))):
    b = os.path.basename(f)
    if os.path.isfile(f) and not b.endswith(".err"):
        shutil.copyfile(f, os.path.join(dst, f"{tag}_{b}"))
with open(os.path.join(dst, f"{tag}_kernel_resources.txt"), "w") as f:
    f.write(subprocess.run(["bash", os.path.join(root, "tools", "kernel_resources.sh")], capture_output=True, text=True).stdout)
for f in sorted(glob.glob(os.path.join(dst, f"{tag}_bench_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:                              # noqa: BLE001
        print(os.path.basename(f), "unreadable:", e)
        continue
    r = d["roofline"]
    print(f"{os.path.basename(f)}: {d['ms_per_step']:.4f} ms/step  {d['value']:.4e} kp+desc/s  one call at a time {d.get('one_batch_ms')}  "
          f"strips {r['launch_ms']:.4f} ms  frac {r['frac']:.4f}  traffic {r.get('traffic')}  valu {(r.get('valu') or {}).get('issue_frac')}  "
          f"stages { {k: round(v, 4) for k, v in r['stage_ms'].items()} }  cpu {d.get('cpu_baseline', {}).get('value')}")
for f in sorted(glob.glob(os.path.join(dst, f"{tag}_kernel_stats_*streams1.csv"))):
    rows = list(csv.DictReader(open(f)))[:7]
    print(os.path.basename(f), [(r["Name"].split("(")[0].split("::")[-1][:24], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1)) for r in rows])
sys.path.insert(0, root)
from pislam_amd import build  # noqa: E402
print("source_hash now", build.source_hash())
# the counters files are only used by bench.py when they were measured on the sources of this tree: say so loudly if not
# (e.g. a file under csrc/ was touched between launching the closing run and its snapshot)
for f in sorted(glob.glob(os.path.join(dst, f"{tag}_counters_*.json"))):
    h = json.load(open(f)).get("source_hash")
    if h != build.source_hash():
        print(f"WARNING: {os.path.basename(f)} was measured on kernel sources {h}, this tree is {build.source_hash()} — "
              f"bench.py will report roofline.traffic / roofline.valu as null: re-run tools/round_final.sh")
