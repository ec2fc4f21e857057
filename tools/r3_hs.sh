#!/bin/bash
# dev: hand-over kernels for the 16-token 1-D windows: parity + bench lines with the switch off / on
python -m pytest tests/test_gpu_modules.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py tests/test_gpu_window_sweep.py tests/test_gpu_padding.py tests/test_gpu_properties.py -x -q -m gpu -k "eva or local or scatter" 2>&1 | tail -3
for rep in 1 2; do for hs in 0 1; do for a in eva; do
EA_WIN_CDIRECT=$hs python bench.py --attn $a --workload cfg5 --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_kernels_avg_us']; print('cdirect=$hs $a cfg5', round(d['ms_per_step'],4), round(d['value']/1e6,1), {n:k[n] for n in k if 'window' in n})"
done; done; done
