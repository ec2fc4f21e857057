#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/gpu_tests17.log 2>&1; echo "pytest rc $?" >> gpurun_out/gpu_tests17.log
run() { # label, env, bench args
  env $2 python bench.py $3 --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$3', round(d['ms_per_step'],4), (d.get('ms_per_step_blocks') or {}).get('median'), {k: v for k, v in d['roofline']['all_kernels_avg_us'].items() if 'linear' in k})"
}
for rep in 1 2; do
  for spec in "--attn lara" "--attn lara --workload cfg2" "--attn eva" "--attn eva --workload cfg2" "--attn softmax" "--attn local --workload cfg2"; do
    run w32 "EA_W192_PREPARE=0" "$spec"
    run w192 "EA_W192_PREPARE=1" "$spec"
  done
done > gpurun_out/ab17.log 2>&1
grep -E "passed|failed|^FAILED|^E  " gpurun_out/gpu_tests17.log | head -20 | cut -c1-250; cat gpurun_out/ab17.log | cut -c1-330
