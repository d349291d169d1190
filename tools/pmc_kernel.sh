#!/bin/bash
# usage: tools/pmc_kernel.sh <outdir-name> [bench args...]  — one PMC pass (SQ counters) over bench.py
cd /tmp && export TMPDIR=/tmp
name=$1; shift
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$name -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1
