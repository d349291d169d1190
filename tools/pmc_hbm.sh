#!/bin/bash
# HBM traffic of the bench step: two separate PMC passes (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2).
# usage: tools/pmc_hbm.sh <outdir-prefix> [bench args...]
cd /tmp && export TMPDIR=/tmp
name=$1; shift
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${name}_fetch -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${name}_write -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${name}_tcc -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1
