#!/bin/bash
# sweep an env knob over bench.py, print the per-kernel times of interest
KNOB=$1; shift
for v in "$@"; do
  env $KNOB=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_kernels_avg_us']
print('$KNOB=$v', 'ms', round(d['ms_per_step'],4), {n:k[n] for n in ('ea_lara_bwd_q_fused','ea_lara_bwd_k_fused','ea_lara_bwd_finish','ea_lara_stats_fwd','ea_lara_out_fwd') if n in k})"
done
