#!/bin/bash
# the GPU suite with each round-3 dev switch OFF: the paths behind them stay reachable for other geometries
for sw in EA_WIN_CDIRECT EA_WIN_HAND_SMALL EA_LM_REG EA_WIN_PLAIN EA_DIRECT_IMPL EA_LARA_COMPOSITE EA_LARA_MODULE_FN EA_PROJ_RS; do
  echo "== $sw=0"; env $sw=0 python -m pytest tests -x -q -m gpu 2>&1 | tail -1
done
