#!/bin/bash
# round 6, GPU call 22: segment kernels: whole-vector LayerNorm arithmetic, head-major unit order (EA_SEGLIN_HMAJOR A/B)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_primitives.py -q -m gpu -x -k "seglin" > gpurun_out/t22a.log 2>&1; echo "rc $?" >> gpurun_out/t22a.log
tail -3 gpurun_out/t22a.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_configs.py -q -m gpu -x -k "lara or cfg5" > gpurun_out/t22b.log 2>&1; echo "rc $?" >> gpurun_out/t22b.log
tail -3 gpurun_out/t22b.log
for hm in 1 0; do
EA_SEGLIN_HMAJOR=$hm bash tools/step_trace.sh gpurun_out/trace22_hm$hm.txt --attn lara --workload cfg5
echo "hmajor $hm"; grep -E "seglin" gpurun_out/trace22_hm$hm.txt | cut -c1-120
done
for rep in 1 2; do
for hm in 1 0; do
EA_SEGLIN_HMAJOR=$hm python bench.py --attn lara --workload cfg5 --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 lara hmajor=$hm', round(d['ms_per_step'],4), (d.get('ms_per_step_blocks') or {}).get('median'), d['value'])"
done
done
