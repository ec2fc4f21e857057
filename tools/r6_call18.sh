#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/gpu_tests18.log 2>&1; echo "pytest rc $?" >> gpurun_out/gpu_tests18.log
tail -3 gpurun_out/gpu_tests18.log
bash tools/switch_matrix.sh > gpurun_out/switch_matrix_r06.txt 2>&1
cat gpurun_out/switch_matrix_r06.txt | cut -c1-200
