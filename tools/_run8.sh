rt=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
mkdir -p gpurun_out/r5h
for opts in "detect_leaks=0:protect_shadow_gap=0:allocator_may_return_null=1" "detect_leaks=0:allocator_may_return_null=1" "detect_leaks=0:protect_shadow_gap=0:allocator_may_return_null=1:max_allocation_size_mb=1"; do
echo "== $opts"
ASAN_OPTIONS=$opts LD_PRELOAD=$rt timeout 120 python -c "
import torch
print('cuda', torch.cuda.is_available())
x = torch.zeros(4, device='cuda'); print(x.sum().item())
" 2>&1 | grep -v amdgpu.ids | head -8
done
echo "== xnack"
HSA_XNACK=1 ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 LD_PRELOAD=$rt timeout 120 python -c "
import torch
x = torch.zeros(4, device='cuda'); print(x.sum().item())
" 2>&1 | grep -v amdgpu.ids | head -8
