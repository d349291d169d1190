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
    -fno-omit-frame-pointer -fsanitize=address -shared-libasan -fno-gpu-sanitize -o $lib $root/pislam_amd/csrc/pislam_hip.hip && echo built $lib ;;
cpu)
  cd $root && LD_PRELOAD=$rt PISLAM_HIP_LIB=$lib python -m pytest tests/test_abi.py -q -m "not gpu" 2>&1 | tail -5 ;;
gpu)
  out=${2:-$root/gpurun_out/asan_gpu.txt}
  cd $root
  {
    echo "# tools/asan_round.sh gpu: libpislam_hip_asan.so (host code under AddressSanitizer), LD_PRELOAD=$rt"
    echo "# ASAN_OPTIONS=$ASAN_OPTIONS ; kernel sources $(python -c 'from pislam_amd import build; print(build.source_hash())' 2>/dev/null)"
    for t in "tests/test_abi.py" "tests/test_gpu_parity.py -k 'not bench and not ranks and not rccl and not exchange and not cpp_'" "tests/test_gpu_fuzz.py" "tests/test_prep.py tests/test_match.py tests/test_tools.py -k 'not cpp'"; do
      echo "## pytest $t -m gpu"
      LD_PRELOAD=$rt PISLAM_HIP_LIB=$lib timeout 1500 bash -c "python -m pytest $t -q -m gpu -x 2>&1 | tail -6"
    done
    echo "## AddressSanitizer reports in the logs above: $(grep -c 'ERROR: AddressSanitizer' $out 2>/dev/null || echo 0)"
  } > $out 2>&1
  tail -30 $out ;;
esac
