#!/bin/bash
# usage (here, no GPU needed): tools/ab_build.sh <name> [-DFLAG ...]  — builds variants/libpislam_hip_<name>.so from the
# working tree with extra compiler flags; on the GPU box `PISLAM_HIP_LIB=variants/libpislam_hip_<name>.so python bench.py ...`
# runs it (development A/B only; variants/ is git-ignored but travels with gpurun).
root=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $root/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -fvisibility=hidden "$@" \
  -o $root/variants/libpislam_hip_$name.so $root/pislam_amd/csrc/pislam_hip.hip && echo built variants/libpislam_hip_$name.so
