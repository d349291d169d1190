cd /tmp && export TMPDIR=/tmp
for a in 1 10 2 0; do
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_a$a -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --ablate $a > /dev/null 2>&1
done
