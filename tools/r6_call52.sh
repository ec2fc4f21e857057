#!/bin/bash
# round 6, GPU call 52: 1-D EVA without the all-false mask: tests, A/B by switch at cfg5
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_modules.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py tests/test_gpu_padding.py tests/test_gpu_properties.py tests/test_gpu_harness.py tests/test_gpu_f32_cores.py -q -m gpu -n 2 -k "eva or harness" > gpurun_out/gpu_tests52.log 2>&1; echo "rc $?" >> gpurun_out/gpu_tests52.log; grep -E "^FAILED|passed|failed|Error" gpurun_out/gpu_tests52.log | tail -8
for sw in 1 0 1 0; do
  EA_EVA_1D_NO_MASK=$sw python bench.py --attn eva --workload cfg5 --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 eva no_mask=$sw', d['ms_per_step'], d.get('ms_per_step_blocks'))"
done 2>&1 | tee gpurun_out/ab52.txt
