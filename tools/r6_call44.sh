#!/bin/bash
# round 6, GPU call 44: causal EVA's training step as one autograd node: tests, then A/B by switch on the LM layer
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_causal_eva.py tests/test_gpu_modules.py tests/test_gpu_harness.py tests/test_gpu_f32_cores.py -q -m gpu -n 2 -k "causal or harness" > gpurun_out/gpu_tests44.log 2>&1; echo "rc $?" >> gpurun_out/gpu_tests44.log; grep -E "^FAILED|passed|failed|Error" gpurun_out/gpu_tests44.log | tail -12
for sw in 1 0 1 0; do
  EA_CAUSAL_MODULE_FN=$sw python bench.py --attn causal_eva --workload lm --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lm module=$sw', d['ms_per_step'], d.get('ms_per_step_blocks'))"
done 2>&1 | tee gpurun_out/ab44.txt
