#!/bin/bash
# round 6, GPU call 6: suite (320-wide wgrad tiles, table bias in EvaAttnFn / causal EVA, paired casts); A/B on the touched workloads
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/gpu_tests6.log 2>&1; echo "pytest rc $?" >> gpurun_out/gpu_tests6.log
run() { # label, env, bench args
  env $2 python bench.py $3 --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$3', round(d['ms_per_step'],4), (d.get('ms_per_step_blocks') or {}).get('median'))"
}
for rep in 1 2; do
  run tile64 "EA_WGRAD_TILE_MAX=255" "--attn eva --batch 32 --grid 24 --dim 320 --heads 5 --window 8 --landmarks 36"
  run tile320 "EA_X=1" "--attn eva --batch 32 --grid 24 --dim 320 --heads 5 --window 8 --landmarks 36"
  run dense "EA_TABLE_BIAS=0" "--attn causal_eva --workload lm"
  run table "EA_TABLE_BIAS=1" "--attn causal_eva --workload lm"
  run lara "EA_X=1" "--attn lara --workload cfg5"
  run lara_b1 "EA_X=1" "--attn lara --workload cfg5 --batch 1"
done > gpurun_out/ab6.log 2>&1
tail -6 gpurun_out/gpu_tests6.log | cut -c1-300; cat gpurun_out/ab6.log
