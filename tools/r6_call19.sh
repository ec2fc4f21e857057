#!/bin/bash
# round 6, GPU call 19: segment kernels of LARA 'adaptive-1d' as one stream per wave (ea_lara_seglin.hip): parity + per-kernel times
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_primitives.py -q -m gpu -x -k "seglin" > gpurun_out/t19a.log 2>&1; echo "rc $?" >> gpurun_out/t19a.log
tail -3 gpurun_out/t19a.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_configs.py tests/test_gpu_modules.py -q -m gpu -x -k "lara" > gpurun_out/t19b.log 2>&1; echo "rc $?" >> gpurun_out/t19b.log
tail -3 gpurun_out/t19b.log
bash tools/step_trace.sh gpurun_out/trace19_cfg5_lara.txt --attn lara --workload cfg5
cat gpurun_out/trace19_cfg5_lara.txt | cut -c1-140
for rep in 1 2; do
python bench.py --attn lara --workload cfg5 --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 lara', round(d['ms_per_step'],4), (d.get('ms_per_step_blocks') or {}).get('median'), d['value'])"
done
