#!/bin/bash
# LDS-side counters of the bench step.  usage: tools/pmc_lds.sh <outdir-name> [bench args...]
cd /tmp && export TMPDIR=/tmp
name=$1; shift
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$name -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/$name.log 2>&1
