#!/bin/bash
# round 6, GPU call 29: ea_wgrad with two stages of rows in flight (tiles up to 192 x 128): parity + PvT stage 3
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_primitives.py -q -m gpu -x -k "wgrad" > gpurun_out/t29.log 2>&1; echo "rc $?" >> gpurun_out/t29.log; tail -3 gpurun_out/t29.log
bash tools/step_trace.sh gpurun_out/trace29_s3.txt --attn eva --batch 32 --dim 320 --heads 5 --grid 24 --window 8 --landmarks 36
grep -E "wgrad|launches" gpurun_out/trace29_s3.txt | cut -c1-130
for rep in 1 2; do
python bench.py --attn eva --batch 32 --dim 320 --heads 5 --grid 24 --window 8 --landmarks 36 --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stage3', round(d['ms_per_step'],4), (d.get('ms_per_step_blocks') or {}).get('median'))"
done
