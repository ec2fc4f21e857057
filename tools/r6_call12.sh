#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { # label, env, bench args
  env $2 python bench.py $3 --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$3', round(d['ms_per_step'],4), (d.get('ms_per_step_blocks') or {}).get('median'), {k: v for k, v in d['roofline']['all_kernels_avg_us'].items() if 'window' in k})"
}
for rep in 1 2; do
  for spec in "--attn eva --workload cfg2" "--attn eva" "--attn local --workload cfg2" "--attn eva --batch 32 --grid 24 --dim 320 --heads 5 --window 8 --landmarks 36" "--attn eva --batch 32 --grid 48 --dim 128 --heads 2 --window 8 --landmarks 36"; do
    run p04 "EA_WIN_BWD_PROLOGUE=0.4" "$spec"
    run p11 "EA_WIN_BWD_PROLOGUE=1.1" "$spec"
  done
done > gpurun_out/ab12.log 2>&1
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/gpu_tests12.log 2>&1; echo "pytest rc $?" >> gpurun_out/gpu_tests12.log
cat gpurun_out/ab12.log | cut -c1-300; tail -3 gpurun_out/gpu_tests12.log
