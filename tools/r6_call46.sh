#!/bin/bash
# round 6, GPU call 46: ScatterBrain on the single-node path: tests, A/B by switch at cfg3
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_modules.py tests/test_gpu_ra.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -q -m gpu -n 2 -k "scatter" > gpurun_out/gpu_tests46.log 2>&1; echo "rc $?" >> gpurun_out/gpu_tests46.log; grep -E "^FAILED|passed|failed|Error" gpurun_out/gpu_tests46.log | tail -12
for sw in 1 0 1 0; do
  EA_GRAPH_CORE=$sw python bench.py --attn scatterbrain --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3 scatterbrain graph_core=$sw', d['ms_per_step'], d.get('ms_per_step_blocks'))"
done 2>&1 | tee gpurun_out/ab46.txt
