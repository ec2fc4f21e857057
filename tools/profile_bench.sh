#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + HBM PMC passes of bench.py.
# usage: tools/profile_bench.sh <attn> <tag>
set -u
ATTN=${1:-lara}; TAG=${2:-r01}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/prof_${TAG}_${ATTN}
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/bench.py --attn $ATTN --steps 10 --warmup 2 --no-graph --no-cpu-baseline > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python $R/bench.py --attn $ATTN --steps 3 --warmup 1 --no-graph --no-cpu-baseline > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python $R/bench.py --attn $ATTN --steps 3 --warmup 1 --no-graph --no-cpu-baseline > $OUT/bench_write.log 2>&1
cd $R
find $OUT -name "*.csv" | head -20
# keep only the small summaries (kernel stats + counter collection); drop the per-dispatch traces if huge
du -sh $OUT
