#!/usr/bin/env python3
"""Run bench.py with the given extra args and print a one-line summary (value, ms/step, stage ms)."""
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline"] + sys.argv[1:], capture_output=True, text=True)
try:
    d = json.loads(out.stdout.strip().splitlines()[-1])
    r = d["roofline"]
    print(" ".join(sys.argv[1:]), "| kp/s %.4g | ms/step %.4f | gpu ms %.4f | stages %s | frac %.4f" % (
        d["value"], d["ms_per_step"], r["step_gpu_ms"], {k: round(v, 4) for k, v in r["stage_ms"].items()}, r["frac"]))
except Exception as e:
    print("FAILED", e, out.stdout[-500:], out.stderr[-1500:])
