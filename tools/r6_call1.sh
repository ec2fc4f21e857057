#!/bin/bash
# round 6, GPU call 1: suite on the no-scratch dgrad_fin + hygiene; A/B of dgrad_fin variants; cfg2 / cfg5 kernel traces
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/gpu_tests.log
bash tools/ab_builds.sh "tools/bin/libea_hip_r05.so efficient-attention_amd/lib/libea_hip.so tools/bin/libea_hip_nopool.so" "lara" "eva" > gpurun_out/ab1.log 2>&1
cd /tmp
for spec in "lara cfg2" "eva cfg2" "lara cfg5" "eva cfg5"; do
  set -- $spec
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tr_$1_$2 -o t -- python $R/bench.py --no-other-workloads --attn $1 --workload $2 --steps 10 --warmup 2 --no-graph --no-cpu-baseline > $R/gpurun_out/tr_$1_$2.log 2>&1
  rm -f $R/gpurun_out/tr_$1_$2/*/*kernel_trace.csv $R/gpurun_out/tr_$1_$2/*kernel_trace.csv
done
cd $R
tail -5 gpurun_out/gpu_tests.log; cat gpurun_out/ab1.log
