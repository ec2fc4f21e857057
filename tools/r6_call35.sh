#!/bin/bash
# round 6, GPU call 35: T5 table gradient of the LM layer -- heads summed by the colsum, long position lists in pieces; A/B by switch
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_causal_eva.py tests/test_gpu_modules.py tests/test_gpu_harness.py -q -m gpu -n 2 -k "table_bias or causal or colsum or harness" > gpurun_out/gpu_tests35.log 2>&1; echo "rc $?" >> gpurun_out/gpu_tests35.log; tail -4 gpurun_out/gpu_tests35.log
for sw in "1 1" "0 0" "1 0" "0 1" "1 1" "0 0"; do
  set -- $sw
  EA_BIAS_HEAD_SUM=$1 EA_TABLE_BIAS_SPLIT=$2 python bench.py --attn causal_eva --workload lm --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lm head_sum=$1 split=$2', d['ms_per_step'], d.get('ms_per_step_blocks'))"
done 2>&1 | tee gpurun_out/ab35.txt
