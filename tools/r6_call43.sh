#!/bin/bash
# round 6, GPU call 43: LinearRA 'adaptive-1d' as one autograd node (GraphCore in CoreModuleFn): tests, then A/B by switch at cfg5
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_modules.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py tests/test_gpu_harness.py -q -m gpu -n 2 -k "lara or harness" > gpurun_out/gpu_tests43.log 2>&1; echo "rc $?" >> gpurun_out/gpu_tests43.log; grep -E "^FAILED|passed|failed|Error" gpurun_out/gpu_tests43.log | tail -12
for sw in 1 0 1 0; do
  for b in 16 1; do
    EA_LARA_1D_MODULE_FN=$sw python bench.py --attn lara --workload cfg5 --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 lara B=$b module_1d=$sw', d['ms_per_step'], d.get('ms_per_step_blocks'))"
  done
done 2>&1 | tee gpurun_out/ab43.txt
