#!/bin/bash
# full GPU suite with the slowest tests listed (run on the GPU box)
mkdir -p gpurun_out
(time python -m pytest tests -m gpu -x -q --durations=8) > gpurun_out/pytest_last.log 2>&1
grep -n "passed\|failed\|Error" gpurun_out/pytest_last.log | tail -5; grep -A10 "slowest" gpurun_out/pytest_last.log | head -14
