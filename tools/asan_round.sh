#!/bin/bash
# usage: bash tools/asan_round.sh [build|cpu|gpu <out.txt>]   — SURVEY §5 "sanitizers" for the PRODUCT's host side.
#   build : variants/libpislam_hip_asan.so = the library with its HOST code under AddressSanitizer (hipcc -fsanitize=address
#           -shared-libasan -fno-gpu-sanitize: plan building, staging, pointer arithmetic, graph bookkeeping of pislam_hip.hip;
#           device code unchanged).  Runs here (no GPU needed).
#   cpu   : tests/test_abi.py (symbol table, error paths without a device) against it, ASan runtime preloaded.
#   gpu   : on the GPU box: tests/test_abi.py + the GPU parity / fuzz suites against it; summary -> <out.txt>.
# ASan options: leak detection off (the Python interpreter and the HIP runtime keep process-lifetime allocations),
# protect_shadow_gap=0 (the HIP runtime maps device-visible memory into the shadow gap).
root=$(cd "$(dirname "$0")/.." && pwd)
rt=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
lib=$root/variants/libpislam_hip_asan.so
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=1:halt_on_error=1
case "${1:-build}" in
build)
  mkdir -p $root/variants
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -fPIC -shared -fvisibility=hidden \
    -fno-omit-frame-pointer -fsanitize=address -shared-libasan -fno-gpu-sanitize -Xarch_host -fsanitize=undefined \
    -Xarch_host -fno-sanitize=vptr,function -Xarch_host -fno-sanitize-recover=undefined \
    -o $lib $root/pislam_amd/csrc/pislam_hip.hip && echo built $lib
  # the same host code with UBSan traps only (no runtime, no interceptors): the variant the GPU box can run — ROCm's ASan
  # runtime intercepts hsa_amd_memory_pool_allocate and cannot allocate device memory on these boxes
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -fPIC -shared -fvisibility=hidden \
    -Xarch_host -fsanitize=undefined -Xarch_host -fsanitize-trap=undefined -Xarch_host -fno-sanitize=vptr,function \
    -o $root/variants/libpislam_hip_ubsan.so $root/pislam_amd/csrc/pislam_hip.hip && echo built $root/variants/libpislam_hip_ubsan.so ;;
cpu)
  cd $root && LD_PRELOAD=$rt PISLAM_HIP_LIB=$lib python -m pytest tests/test_abi.py -q -m "not gpu" 2>&1 | tail -5
  cd $root && LD_PRELOAD=$rt PISLAM_HIP_LIB=$lib python tests/plan_fuzz.py ${2:-20000} ;;
gpu)
  # ROCm's ASan runtime cannot coexist with the HIP runtime on the GPU boxes (its hsa_amd_memory_pool_allocate interceptor
  # fails: "AddressSanitizer: out of memory", profiles/r05_asan_gpu.txt): the GPU suites run against the UBSan-trap build
  # of the host code instead — any undefined behaviour (signed overflow, invalid shift, misaligned or null access,
  # out-of-bounds index of a fixed array) in plan building, staging or graph bookkeeping kills the test process.
  out=${2:-$root/gpurun_out/ubsan_gpu.txt}
  cd $root
  {
    echo "# tools/asan_round.sh gpu: variants/libpislam_hip_ubsan.so (host code: -fsanitize=undefined -fsanitize-trap=undefined), kernel sources $(python -c 'from pislam_amd import build; print(build.source_hash())' 2>/dev/null)"
    echo "## ASan runtime + HIP runtime on this box:"
    ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 LD_PRELOAD=$rt timeout 120 python -c "import torch; torch.zeros(4, device='cuda')" 2>&1 | grep -v amdgpu.ids | head -4
    for t in "tests/test_abi.py tests/test_gpu_parity.py" "tests/test_gpu_fuzz.py tests/test_prep.py tests/test_match.py tests/test_tools.py"; do
      echo "## PISLAM_HIP_LIB=variants/libpislam_hip_ubsan.so pytest $t -m gpu"
      # (the C++ drop-in / tool tests link -lpislam_hip from the in-tree lib directory: not this variant)
      PISLAM_HIP_LIB=$root/variants/libpislam_hip_ubsan.so timeout 1500 python -m pytest $t -q -m gpu -k "not cpp" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" | tail -4
    done
    echo "## tests/fuzz_campaign.py --seeds 3000 --wide"
    PISLAM_HIP_LIB=$root/variants/libpislam_hip_ubsan.so timeout 900 python tests/fuzz_campaign.py --seeds 3000 --wide --start 3500000 2>&1 | tail -1
  } > $out 2>&1
  tail -30 $out ;;
esac
