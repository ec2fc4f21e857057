#!/bin/bash
# round 4 dev: targeted parity tests + bench lines (run on the GPU box)
#   tools/r4_check.sh "<pytest -k expr>" [bench arg sets ...]   -> gpurun_out/r4_check/
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r4_check; mkdir -p $OUT
K=$1; shift
if [ -n "$K" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q -k "$K" > $OUT/pytest.log 2>&1
  tail -5 $OUT/pytest.log
fi
i=0
for args in "$@"; do
  i=$((i+1))
  timeout 600 python bench.py $args > $OUT/bench_$i.log 2>&1
  echo "== bench $args"; tail -1 $OUT/bench_$i.log | cut -c1-1500
done
