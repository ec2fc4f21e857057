#!/bin/bash
# round 6, GPU call 30: the workload x attention matrix with the final library; the suite under the switches next to this round's
# last changes (the folded 'adaptive-1d' paths the segment kernels replace)
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/bench_matrix.sh > gpurun_out/matrix30.txt 2>&1
python bench.py --attn lara --workload cfg5 --batch 1 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg5 lara B=1', round(d['value']/1e6,1), round(d['ms_per_step'],3))" >> gpurun_out/matrix30.txt
cat gpurun_out/matrix30.txt
for sw in EA_SEGLIN=0 "EA_SEGLIN=0 EA_FOLD_KERNELS=0" EA_LARA_FOLD=0; do
  echo "== $sw"
  env $sw timeout 1200 python -m pytest tests -m gpu -q -n 2 2>&1 | grep -E "^FAILED|passed|failed" | tail -8
done > gpurun_out/switch30.txt 2>&1
cat gpurun_out/switch30.txt
