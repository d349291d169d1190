#!/usr/bin/env python3
"""usage: tools/counters_json.py <out.json> <workload> <batch> "<command>" <pass dir> [<pass dir> ...]
Reduces rocprofv3 --pmc passes (each <pass dir>/p_counter_collection.csv: one row per dispatch and counter) to one
JSON per workload — what bench.py quotes as `roofline.traffic` / `roofline.valu` when the kernel sources match:
per kernel the mean of every counter over its dispatches, the launches per step, and
  hbm_bytes_per_launch = 2 * FETCH_SIZE[KiB] * 1024 + WRITE_SIZE[KiB] * 1024
(FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream on gfx950 — MI355X_MICROARCH.md, HBM
section — hence the factor 2; WRITE_SIZE is uncorrected)."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
out, workload, batch, command = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
steps = None
for d in sys.argv[5:]:
    path = os.path.join(d, "p_counter_collection.csv")
    if not os.path.exists(path):
        print("missing", path, file=sys.stderr)
        continue
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if not any(ns in name for ns in ("pf::", "pp::", "pm::", "pk::")):
            continue
        k = name.split("(")[0].split("::")[-1].split("<")[0]
        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
kern = {}
for k, d in vals.items():
    e = {c: sum(v) / len(v) for c, v in d.items()}
    e["dispatches_profiled"] = max(len(v) for v in d.values())
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["read_bytes_corrected"] = 2 * e["FETCH_SIZE"] * 1024
        e["write_bytes"] = e["WRITE_SIZE"] * 1024
        e["hbm_bytes_per_launch"] = e["read_bytes_corrected"] + e["write_bytes"]
    if e.get("TCC_HIT_sum") is not None and e.get("TCC_MISS_sum") is not None and e["TCC_HIT_sum"] + e["TCC_MISS_sum"] > 0:
        e["l2_hit_rate"] = e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"])
    kern[k] = e
# launches per step: relative to the strip kernel (one launch per step)
ref = kern.get("k_fused_strips", {}).get("dispatches_profiled") or 1
for e in kern.values():
    e["launches_per_step"] = round(e["dispatches_profiled"] / ref, 3)
from pislam_amd import build as _b  # noqa: E402
json.dump({"source_hash": _b.source_hash(), "workload": workload, "batch": batch, "command": command,
           "note": "means over the dispatches of each kernel; read bytes = 2 * FETCH_SIZE(KiB) * 1024 (gfx950 correction, "
                   "MI355X_MICROARCH.md HBM section), WRITE_SIZE (KiB) uncorrected; separate rocprofv3 --pmc passes "
                   "(--kernel-trace only beside them)",
           "kernels": kern}, open(out, "w"), indent=1)
for k, e in kern.items():
    print(workload, k, {c: (round(v / 1e6, 2) if isinstance(v, float) and v > 1e4 else v) for c, v in e.items()
                        if c in ("hbm_bytes_per_launch", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "launches_per_step", "l2_hit_rate")})
