#!/bin/bash
# round 6, GPU call 7: suite; LM with / without the table bias; default bench with the clock pre-warm
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/gpu_tests7.log 2>&1; echo "pytest rc $?" >> gpurun_out/gpu_tests7.log
run() { # label, env, bench args
  env $2 python bench.py $3 --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$3', round(d['ms_per_step'],4), (d.get('ms_per_step_blocks') or {}))"
}
for rep in 1 2; do
  run dense "EA_TABLE_BIAS=0" "--attn causal_eva --workload lm"
  run table "EA_TABLE_BIAS=1" "--attn causal_eva --workload lm"
  run nowarm "EA_BENCH_PREWARM_MS=0" "--attn lara"
  run prewarm "EA_BENCH_PREWARM_MS=50" "--attn lara"
done > gpurun_out/ab7.log 2>&1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench7_default.json 2> gpurun_out/bench7_default.err
tail -4 gpurun_out/gpu_tests7.log | cut -c1-300; cat gpurun_out/ab7.log
