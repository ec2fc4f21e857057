#!/bin/bash
# the GPU suite once per dev switch set to its non-default value: the paths behind the switches stay reachable
# (other geometries / tracing take them) and are kept green
for sw in EA_LARA_FOLD=0 EA_DGRAD_RS=0 EA_SEGLIN=0 "EA_SEGLIN=0 EA_FOLD_KERNELS=0" EA_PROJ_POOL=0 EA_MULTI_SUM=0 EA_PERFORMER_16BIT=1 EA_WGRAD_PAIR=0 EA_EVA_COMPOSITE=0 EA_LARA_COMPOSITE=0 "EA_EVA_MODULE_FN=0 EA_LARA_MODULE_FN=0" EA_CORE_MODULE_FN=0 EA_WIN_DBD=0 EA_DGRAD_FIN=0 EA_F32_CORES=0 EA_TABLE_BIAS=0 EA_EVA_FOLD_SLICES=0 EA_WIDE_MODULE_FN=0 EA_DGF_CELLS=0 EA_WIN_BWD_PROLOGUE=0.4 EA_W192_PREPARE=0 EA_SEGLIN_FIN=0 EA_COLSUM_TWO_STAGE=0 EA_TABLE_BIAS_SPLIT=0 EA_BIAS_HEAD_SUM=0 EA_STACKED_LINEAR=0 EA_LARA_1D_MODULE_FN=0 EA_CAUSAL_MODULE_FN=0 EA_GRAPH_CORE=0; do
  echo "== $sw"
  env $sw python -m pytest tests -m gpu -q 2>&1 | grep -E "^FAILED|passed|failed" | tail -8
done
