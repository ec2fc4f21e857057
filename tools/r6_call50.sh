#!/bin/bash
# round 6, GPU call 50: the driver's own command -- the GPU suite sequentially (no xdist), -x
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1150 python -m pytest tests/ -x -q -m gpu > gpurun_out/gpu_tests50.log 2>&1; echo "rc $?" >> gpurun_out/gpu_tests50.log; grep -E "^FAILED|passed|failed|rc " gpurun_out/gpu_tests50.log | tail -5
