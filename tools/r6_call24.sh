#!/bin/bash
# round 6, GPU call 24: beta backward of long overlapping chunks with eight waves per chunk (EA_BETA_BWD_W8 A/B)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu -x -n 2 -k "eva or local or golden" > gpurun_out/t24.log 2>&1; echo "rc $?" >> gpurun_out/t24.log
tail -3 gpurun_out/t24.log
for w in 1 0; do
EA_BETA_BWD_W8=$w bash tools/step_trace.sh gpurun_out/trace24_w$w.txt --attn eva --workload cfg5
echo "w8 $w"; grep -E "beta_bwd|launches" gpurun_out/trace24_w$w.txt | cut -c1-130
done
for rep in 1 2; do
for w in 1 0; do
EA_BETA_BWD_W8=$w python bench.py --attn eva --workload cfg5 --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 eva w8=$w', round(d['ms_per_step'],4), (d.get('ms_per_step_blocks') or {}).get('median'), d['value'])"
done
done
