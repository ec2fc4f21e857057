#!/bin/bash
# dev: softmax parity + bench lines per query-tile count (EA_SM_QT) at cfg3 / cfg5 / cfg2
python -m pytest tests/test_gpu_modules.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py tests/test_gpu_ra.py tests/test_gpu_properties.py tests/test_gpu_padding.py -x -q -m gpu -k "softmax or ra or sample" 2>&1 | tail -3
for qt in ${1:-0}; do
for w in cfg3 cfg5 cfg2; do
EA_SM_QT=$qt python bench.py --attn softmax --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_kernels_avg_us']
print('qt=$qt $w softmax', 'ms', round(d['ms_per_step'],4), 'Mtok/s', round(d['value']/1e6,1), {n:k[n] for n in k if 'softmax' in n})"
done; done
