#!/bin/bash
# usage: tools/kernel_resources.sh <out.txt>  — per-kernel VGPR / SGPR / spills / scratch / occupancy / static LDS as the
# compiler reports them (hipcc -Rpass-analysis=kernel-resource-usage, device pass only; no GPU needed).
# Dynamic LDS (the strip kernels' tiles) is added at launch: see build_fused_plan in pislam_hip.hip.
root=$(cd "$(dirname "$0")/.." && pwd)
out=${1:-/dev/stdout}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -c \
  -Rpass-analysis=kernel-resource-usage "$root/pislam_amd/csrc/pislam_hip.hip" -o /dev/null 2>&1 |
python3 -c '
import re, subprocess, sys
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"remark: (?:\s*)([A-Za-z \[\]/]+): (.*?) \[-Rpass", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        name = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name)}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
cols = ["VGPRs", "AGPRs", "TotalSGPRs", "SGPRs Spill", "VGPRs Spill", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"]
print("kernel | " + " | ".join(cols))
for r in rows:
    print(r["name"] + " | " + " | ".join(r.get(c, "?") for c in cols))
' > "$out"
