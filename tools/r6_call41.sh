#!/bin/bash
# round 6, GPU call 41: the switch-aware versions of the three new primitive tests under their switches (call 40 ran the versions before)
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/switch41.txt
for sw in EA_COLSUM_TWO_STAGE=0 EA_TABLE_BIAS_SPLIT=0 EA_STACKED_LINEAR=0; do
  echo "== $sw" >> gpurun_out/switch41.txt
  env $sw timeout 600 python -m pytest tests/test_gpu_primitives.py -m gpu -q -k "colsum or table_bias or stacked" 2>&1 | grep -E "^FAILED|passed|failed|skipped" | tail -5 >> gpurun_out/switch41.txt
done
cat gpurun_out/switch41.txt
