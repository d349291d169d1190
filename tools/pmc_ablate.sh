#!/bin/bash
# VALU / LDS instruction counts of the strip kernel per ablation level (profiling only).
cd /tmp && export TMPDIR=/tmp
for a in 1 26 10 12 8 0; do
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_a$a -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --ablate $a > /dev/null 2>&1
done
