#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/ab_builds.sh "tools/bin/libea_hip_base.so efficient-attention_amd/lib/libea_hip.so" "lara --workload cfg2" "lara" > gpurun_out/ab11.log 2>&1
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_modules.py tests/test_gpu_configs.py tests/test_gpu_primitives.py -q -m gpu -x > gpurun_out/gpu_tests11.log 2>&1; echo "pytest rc $?" >> gpurun_out/gpu_tests11.log
grep -E "^(lara|eva)" gpurun_out/ab11.log | cut -c1-400; tail -3 gpurun_out/gpu_tests11.log
