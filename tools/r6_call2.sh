#!/bin/bash
# round 6, GPU call 2: the whole suite (no -x) on the trash-line fix + TableBias; default bench
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/gpu_tests2.log 2>&1; echo "pytest rc $?" >> gpurun_out/gpu_tests2.log
timeout 600 python bench.py > gpurun_out/bench2_default.json 2> gpurun_out/bench2_default.err
tail -15 gpurun_out/gpu_tests2.log | cut -c1-300
