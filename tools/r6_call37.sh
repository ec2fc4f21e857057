#!/bin/bash
# round 6, GPU call 37: StackedLinearFn (no fp32 concatenation of causal EVA's q / k / v weights) and the bench's multi-tensor SGD
# for many large parameters; tests, then A/B by switch on the LM layer
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_causal_eva.py tests/test_gpu_modules.py tests/test_gpu_harness.py tests/test_gpu_f32_cores.py -q -m gpu -n 2 -k "stacked or causal or harness" > gpurun_out/gpu_tests37.log 2>&1; echo "rc $?" >> gpurun_out/gpu_tests37.log; tail -4 gpurun_out/gpu_tests37.log
for sw in "1 1" "0 0" "1 0" "0 1" "1 1" "0 0"; do
  set -- $sw
  EA_STACKED_LINEAR=$1 EA_BENCH_SGD_FOREACH_BIG=$2 python bench.py --attn causal_eva --workload lm --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lm stacked=$1 sgd_foreach=$2', d['ms_per_step'], d.get('ms_per_step_blocks'))"
done 2>&1 | tee gpurun_out/ab37.txt
