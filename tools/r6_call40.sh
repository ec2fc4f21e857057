#!/bin/bash
# round 6, GPU call 40: the causal-EVA / primitive / module / harness / full-size tests under the four switches of the last session
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/switch40.txt
for sw in EA_COLSUM_TWO_STAGE=0 EA_TABLE_BIAS_SPLIT=0 EA_BIAS_HEAD_SUM=0 EA_STACKED_LINEAR=0; do
  echo "== $sw" >> gpurun_out/switch40.txt
  env $sw timeout 900 python -m pytest tests/test_gpu_causal_eva.py tests/test_gpu_primitives.py tests/test_gpu_modules.py tests/test_gpu_harness.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -m gpu -q -n 2 2>&1 | grep -E "^FAILED|passed|failed" | tail -8 >> gpurun_out/switch40.txt
done
cat gpurun_out/switch40.txt
