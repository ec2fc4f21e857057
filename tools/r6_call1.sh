#!/bin/bash
# round 6, GPU call 1: suite on the no-scratch library + hygiene; default bench; cfg2 / cfg5 kernel traces
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/gpu_tests.log
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
timeout 300 python bench.py --attn eva --no-cpu-baseline --no-other-workloads > gpurun_out/bench_eva.json 2>/dev/null
cd /tmp
for spec in "lara cfg3" "eva cfg3" "lara cfg2" "eva cfg2" "lara cfg5" "eva cfg5"; do
  set -- $spec
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tr_$1_$2 -o t -- python $R/bench.py --no-other-workloads --attn $1 --workload $2 --steps 10 --warmup 2 --no-graph --no-cpu-baseline > $R/gpurun_out/tr_$1_$2.log 2>&1
  rm -f $R/gpurun_out/tr_$1_$2/*/*kernel_trace.csv $R/gpurun_out/tr_$1_$2/*kernel_trace.csv
done
cd $R
tail -5 gpurun_out/gpu_tests.log; cat gpurun_out/bench_default.json | cut -c1-1500
