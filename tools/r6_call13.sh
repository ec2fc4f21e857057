#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { # label, env, bench args
  env $2 python bench.py $3 --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$3', round(d['ms_per_step'],4), (d.get('ms_per_step_blocks') or {}).get('median'), {k: v for k, v in d['roofline']['all_kernels_avg_us'].items() if 'window' in k})"
}
for rep in 1 2; do
  for spec in "--attn eva" "--attn eva --workload cfg2" "--attn local" "--attn eva --batch 32 --grid 96 --dim 64 --heads 1 --window 8 --landmarks 36" "--attn eva --batch 32 --grid 48 --dim 128 --heads 2 --window 8 --landmarks 36" "--attn eva --workload cfg5"; do
    for v in 0.4 0.8 1.2 2.0; do
      run f$v "EA_WIN_FWD_PROLOGUE=$v" "$spec"
    done
  done
done > gpurun_out/ab13.log 2>&1
cat gpurun_out/ab13.log | cut -c1-300
