#!/bin/bash
# round 6, GPU call 3: suite on the slice-sum fold; A/B of the two round-6 switches on EVA cfg3 / cfg2
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/gpu_tests3.log 2>&1; echo "pytest rc $?" >> gpurun_out/gpu_tests3.log
run() { # label, env, bench args
  env $2 python bench.py $3 --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$3', round(d['ms_per_step'],4), (d.get('ms_per_step_blocks') or {}).get('median'))"
}
for rep in 1 2; do
  for wl in cfg3 cfg2; do
    run base "EA_EVA_FOLD_SLICES=0 EA_TABLE_BIAS=0" "--attn eva --workload $wl"
    run table "EA_EVA_FOLD_SLICES=0 EA_TABLE_BIAS=1" "--attn eva --workload $wl"
    run both "EA_EVA_FOLD_SLICES=1 EA_TABLE_BIAS=1" "--attn eva --workload $wl"
  done
  run base "EA_TABLE_BIAS=0" "--attn local --workload cfg3"
  run table "EA_TABLE_BIAS=1" "--attn local --workload cfg3"
done > gpurun_out/ab3.log 2>&1
tail -4 gpurun_out/gpu_tests3.log | cut -c1-200; cat gpurun_out/ab3.log
