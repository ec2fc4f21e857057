#!/bin/bash
# dev: LinearRA as one autograd node: parity + eager / captured step with the switch off / on
python -m pytest tests/test_gpu_modules.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py tests/test_gpu_custom_ops.py tests/test_gpu_padding.py tests/test_gpu_primitives.py tests/test_gpu_properties.py tests/test_gpu_harness.py tests/test_gpu_bench_ddp.py -x -q -m gpu -k "lara or harness or ddp or compiled" 2>&1 | tail -2
for m in 0 1 0 1; do EA_LARA_MODULE_FN=$m python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('module_fn=$m', round(d['ms_per_step'],4), 'eager', d.get('eager_ms_per_step'), d.get('eager_ms_per_step_blocks'))"; done
