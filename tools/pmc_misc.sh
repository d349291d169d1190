#!/bin/bash
# Issue-side counters of the bench step.  usage: tools/pmc_misc.sh <outdir-name> [bench args...]
cd /tmp && export TMPDIR=/tmp
name=$1; shift
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SMEM --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$name -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/$name.log 2>&1
