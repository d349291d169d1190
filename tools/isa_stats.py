#!/usr/bin/env python3
"""Static instruction statistics of the device code (no GPU needed):
   tools/isa_stats.py [kernel-name-substring ...]   — compiles pislam_hip.hip to gfx950 assembly and counts, per
kernel, VALU / SALU / DS / VMEM instructions, SGPR-spill traffic (v_readlane / v_writelane), waits, branches."""
import collections, os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
asm = "/tmp/pislam_dev.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                       "--cuda-device-only", "-S", os.path.join(root, "pislam_amd/csrc/pislam_hip.hip"), "-o", asm],
                      stderr=subprocess.DEVNULL)
pats = sys.argv[1:] or ["k_fused_stripsILb1ELb0ELb1", "k_gather_orb", "k_gather_descE"]
cur, body = None, collections.defaultdict(list)
for line in open(asm):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur = m.group(1)
        continue
    if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
        cur = None
    if cur and line.startswith("\t") and not line.strip().startswith((".", ";")):
        body[cur].append(line.strip().split()[0])
for name, ins in body.items():
    if not any(p in name for p in pats):
        continue
    c = collections.Counter(ins)
    grp = lambda pre: sum(n for k, n in c.items() if k.startswith(pre))
    print(f"{name[:70]}: total {len(ins)} valu {grp('v_')} salu {grp('s_')} ds {grp('ds_')} vmem {grp('global_') + grp('buffer_') + grp('flat_')}")
    print(f"   readlane {c['v_readlane_b32']} writelane {c['v_writelane_b32']} readfirstlane {c['v_readfirstlane_b32']} "
          f"s_load {grp('s_load')} waitcnt {c['s_waitcnt']} branch {grp('s_cbranch') + c['s_branch']} barrier {c['s_barrier']} "
          f"flat {grp('flat_')} scratch {grp('scratch_')}")
