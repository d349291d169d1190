#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel (millions) from gpurun_out/<name>/p_counter_collection.csv."""
import collections, csv, sys
for name in sys.argv[1:]:
    rows = list(csv.DictReader(open(f"gpurun_out/{name}/p_counter_collection.csv")))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(collections.Counter)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0][-40:]
        if "pf::" not in r["Kernel_Name"] and "pk::" not in r["Kernel_Name"]:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
    for k in agg:
        print(name, k, {c: round(v / n[k][c] / 1e6, 2) for c, v in sorted(agg[k].items())})
