#!/bin/bash
# round 6, GPU call 23: EVA landmark streaming loops with U row steps per round trip: full GPU suite + cfg5 EVA / LM traces
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/step_trace.sh gpurun_out/trace23_cfg5_eva.txt --attn eva --workload cfg5
grep -E "beta|chunk_mean|launches" gpurun_out/trace23_cfg5_eva.txt | cut -c1-130
bash tools/step_trace.sh gpurun_out/trace23_lm.txt --attn causal_eva --workload lm
grep -E "beta|chunk_mean|launches" gpurun_out/trace23_lm.txt | cut -c1-130
for rep in 1 2; do
python bench.py --attn eva --workload cfg5 --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 eva', round(d['ms_per_step'],4), (d.get('ms_per_step_blocks') or {}).get('median'), d['value'])"
done
timeout 1700 python -m pytest tests -q -m gpu -x -n 2 > gpurun_out/gpu_tests23.log 2>&1; echo "pytest rc $?" >> gpurun_out/gpu_tests23.log
tail -4 gpurun_out/gpu_tests23.log
