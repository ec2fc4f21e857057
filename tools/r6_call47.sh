#!/bin/bash
# round 6, GPU call 47: the ScatterBrain single-node test (call 46 hit a syntax error in the test file)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_primitives.py -q -m gpu -k "scatter or randomized or lara_1d or stacked" > gpurun_out/gpu_tests47.log 2>&1; echo "rc $?" >> gpurun_out/gpu_tests47.log; grep -E "^FAILED|passed|failed|Error|assert" gpurun_out/gpu_tests47.log | tail -12
