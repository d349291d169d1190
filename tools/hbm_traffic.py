#!/usr/bin/env python3
"""HBM bytes per launch from the three PMC passes of tools/pmc_hbm.sh -> profiles/<out>.json.

usage: tools/hbm_traffic.py <gpurun_out prefix> <out json> [note]
FETCH_SIZE is reported in KiB and on gfx950 counts 64 B per 128-B request of a wide coalesced stream
(MI355X_MICROARCH.md, HBM section): read bytes = 2 * FETCH_SIZE * 1024.  WRITE_SIZE (KiB) is uncorrected.
"""
import collections, csv, json, os, sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def per_kernel(path):
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(collections.Counter)
    for r in rows:
        name = r["Kernel_Name"]
        if "pf::" not in name and "pp::" not in name and "pm::" not in name:
            continue
        k = name.split("(")[0].split("::")[-1].split("<")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k][r["Counter_Name"]] += 1
    return {k: {c: v / n[k][c] for c, v in d.items()} for k, d in agg.items()}


def main():
    prefix, out = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    kern = collections.defaultdict(dict)
    for part in ("fetch", "write", "tcc"):
        for k, d in per_kernel(f"gpurun_out/{prefix}_{part}/p_counter_collection.csv").items():
            kern[k].update(d)
    for k, d in kern.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["read_bytes_corrected"] = 2 * d["FETCH_SIZE"] * 1024
            d["write_bytes"] = d["WRITE_SIZE"] * 1024
            d["hbm_bytes_per_launch"] = d["read_bytes_corrected"] + d["write_bytes"]
        if d.get("TCC_HIT_sum") is not None and d.get("TCC_MISS_sum") is not None and d["TCC_HIT_sum"] + d["TCC_MISS_sum"] > 0:
            d["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
    from pislam_amd import build as _b
    json.dump({"source_hash": _b.source_hash(),
               "source": "rocprofv3 --pmc, separate passes (FETCH_SIZE | WRITE_SIZE | TCC_*), tools/pmc_hbm.sh + "
                         "tools/hbm_traffic.py; bench.py --steps 3 --warmup 1 (batch 256, synthetic VGA pyramids), MI355X. " + note,
               "note": "read bytes = 2 * FETCH_SIZE(KiB) * 1024 (gfx950 correction, MI355X_MICROARCH.md HBM section); "
                       "WRITE_SIZE (KiB) uncorrected; averages over the launches of each kernel",
               "kernels": kern}, open(out, "w"), indent=1)
    for k, d in kern.items():
        print(k, {c: round(v, 1) for c, v in d.items() if c in ("hbm_bytes_per_launch", "l2_hit_rate")})


if __name__ == "__main__":
    main()
