#!/usr/bin/env python3
"""usage: tools/pmc_aggregate.py <p_counter_collection.csv> <out.csv>
Reduces a rocprofv3 --pmc counter collection (one row per dispatch and counter, megabytes for a bench run) to
one row per (pf:: kernel, counter): dispatches, mean / min / max of the counter, and the dispatch's resource
columns as rocprofv3 reports them — the form profiles/ keeps."""
import collections
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
vals = collections.defaultdict(list)
meta = {}
for r in csv.DictReader(open(src)):
    if "pf::" not in r["Kernel_Name"] and "pp::" not in r["Kernel_Name"]:
        continue
    k = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
    vals[k].append(float(r["Counter_Value"]))
    meta[k] = r
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Kernel_Name", "Counter_Name", "dispatches", "mean", "min", "max", "VGPR_Count(rocprof granules)",
                "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size(static)", "Scratch_Size", "Workgroup_Size", "Grid_Size"])
    for k in sorted(vals):
        v, r = vals[k], meta[k]
        w.writerow([k[0], k[1], len(v), sum(v) / len(v), min(v), max(v), r.get("VGPR_Count"), r.get("Accum_VGPR_Count"),
                    r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size"), r.get("Workgroup_Size"), r.get("Grid_Size")])
