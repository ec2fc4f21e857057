#!/bin/bash
# round 6, GPU call 4: suite on the wide single-node path; A/B of EA_WIDE_MODULE_FN on the wide workloads
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/gpu_tests4.log 2>&1; echo "pytest rc $?" >> gpurun_out/gpu_tests4.log
run() { # label, env, bench args
  env $2 python bench.py $3 --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$3', round(d['ms_per_step'],4), (d.get('ms_per_step_blocks') or {}).get('median'))"
}
for rep in 1 2; do
  for spec in "--attn eva --batch 32 --grid 24 --dim 320 --heads 5 --window 8 --landmarks 36" "--attn softmax --batch 32 --grid 12 --dim 512 --heads 8" "--attn softmax --workload cfg5" "--attn eva --workload cfg5" "--attn local --workload cfg5"; do
    run three "EA_WIDE_MODULE_FN=0" "$spec"
    run one "EA_WIDE_MODULE_FN=1" "$spec"
  done
done > gpurun_out/ab4.log 2>&1
tail -6 gpurun_out/gpu_tests4.log | cut -c1-300; cat gpurun_out/ab4.log
