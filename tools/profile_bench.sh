#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + HBM PMC passes of bench.py.
# usage: tools/profile_bench.sh <attn> <tag> [extra bench.py flags, e.g. "--workload lm"]
set -u
ATTN=${1:-lara}; TAG=${2:-r01}; EXTRA=${3:-}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/prof_${TAG}_${ATTN}
rm -rf $OUT; mkdir -p $OUT
cd /tmp
# GEMM solutions are selected once, outside the profiler (hundreds of trial kernels), and reused
TUNE=$OUT/tunableop.csv
python $R/bench.py --no-other-workloads --attn $ATTN $EXTRA --steps 2 --warmup 2 --no-graph --no-cpu-baseline --gemm-tune-file $TUNE > $OUT/bench_tune.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/bench.py --no-other-workloads --attn $ATTN $EXTRA --steps 10 --warmup 2 --no-graph --no-cpu-baseline --gemm-tune-file $TUNE > $OUT/bench_trace.log 2>&1
# counters: separate passes, no tracing options
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python $R/bench.py --no-other-workloads --attn $ATTN $EXTRA --steps 3 --warmup 1 --no-graph --no-cpu-baseline --gemm-tune-file $TUNE > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python $R/bench.py --no-other-workloads --attn $ATTN $EXTRA --steps 3 --warmup 1 --no-graph --no-cpu-baseline --gemm-tune-file $TUNE > $OUT/bench_write.log 2>&1
cd $R
# keep the summaries small: per-dispatch rows only for this library's kernels
for f in $OUT/pmc_fetch/f_counter_collection.csv $OUT/pmc_write/w_counter_collection.csv $OUT/trace/t_kernel_trace.csv; do
  if [ -f $f ]; then (head -1 $f; grep "ea::" $f) > $f.tmp; mv $f.tmp $f; fi
done
head -40 $OUT/trace/t_kernel_stats.csv | cut -c1-160
du -sh $OUT
