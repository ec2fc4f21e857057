#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/gpu_tests15.log 2>&1; echo "pytest rc $?" >> gpurun_out/gpu_tests15.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench15_default.json 2> gpurun_out/bench15_default.err
timeout 300 python bench.py --attn eva --steps 20 --warmup 5 --no-other-workloads --no-cpu-baseline > gpurun_out/bench15_eva.json 2>/dev/null
tail -3 gpurun_out/gpu_tests15.log; cut -c1-600 gpurun_out/bench15_default.json
