#!/bin/bash
# round 6, GPU call 9: A/B of library builds (base = commit 5c606f5, new = the w_cast pieces spread over the workgroups)
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/ab_builds.sh "tools/bin/libea_hip_base.so efficient-attention_amd/lib/libea_hip.so" "lara" "lara --workload cfg2" "eva --workload cfg2" > gpurun_out/ab9.log 2>&1
timeout 600 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_modules.py -q -m gpu -x > gpurun_out/gpu_tests9.log 2>&1; echo "pytest rc $?" >> gpurun_out/gpu_tests9.log
cat gpurun_out/ab9.log | cut -c1-400; tail -3 gpurun_out/gpu_tests9.log
