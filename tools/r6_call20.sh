#!/bin/bash
# round 6, GPU call 20: ea_lara_seglin_bwd_fin (the estimator's last dq correction inside the segment backward): parity + times
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_primitives.py -q -m gpu -x -k "seglin" > gpurun_out/t20a.log 2>&1; echo "rc $?" >> gpurun_out/t20a.log
tail -5 gpurun_out/t20a.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_configs.py tests/test_gpu_modules.py tests/test_gpu_harness.py -q -m gpu -x -k "lara or cfg5" > gpurun_out/t20b.log 2>&1; echo "rc $?" >> gpurun_out/t20b.log
tail -3 gpurun_out/t20b.log
bash tools/step_trace.sh gpurun_out/trace20_cfg5_lara.txt --attn lara --workload cfg5
grep -E "seglin|fin_kernel|launches" gpurun_out/trace20_cfg5_lara.txt | cut -c1-140
for rep in 1 2; do
for sw in 1 0; do
EA_SEGLIN_FIN=$sw python bench.py --attn lara --workload cfg5 --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 lara fin=$sw', round(d['ms_per_step'],4), (d.get('ms_per_step_blocks') or {}).get('median'), d['value'])"
done
done
EA_SEGLIN_FIN=1 python bench.py --attn lara --workload cfg5 --batch 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 lara B=1', round(d['ms_per_step'],4), (d.get('ms_per_step_blocks') or {}).get('median'), d['value'])"
