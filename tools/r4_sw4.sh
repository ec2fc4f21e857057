#!/bin/bash
# dev: the fast GPU test files under the switch sets that change module-level routing
for sw in "EA_EVA_MODULE_FN=0 EA_LARA_MODULE_FN=0" EA_PROJ_POOL=0; do
  echo "== $sw"
  env $sw python -m pytest tests/test_gpu_primitives.py tests/test_gpu_modules.py tests/test_gpu_configs.py -m gpu -q 2>&1 | grep -E "^FAILED|passed|failed" | tail -8
done
