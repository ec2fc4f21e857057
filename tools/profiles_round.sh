#!/bin/bash
# all committed evidence of a round in one GPU call: kernel-trace stats + HBM PMC (profile_bench.sh) and SQ/MFMA counters (pmc_sq.sh)
TAG=${1:-r03}
for a in lara eva; do bash tools/profile_bench.sh $a $TAG > gpurun_out/prof_${TAG}_$a.log 2>&1; done
bash tools/profile_bench.sh softmax $TAG "--no-other-workloads" > gpurun_out/prof_${TAG}_softmax.log 2>&1
for a in lara eva softmax; do bash tools/pmc_sq.sh $a > gpurun_out/sq_$a.log 2>&1; done
# the softmax baseline at N = 4096 (cfg5): kernel durations + SQ / MFMA counters
bash tools/profile_bench.sh softmax ${TAG}cfg5 "--workload cfg5" > gpurun_out/prof_${TAG}cfg5_softmax.log 2>&1
bash tools/pmc_sq.sh softmax cfg5 > gpurun_out/sq_softmax_cfg5.log 2>&1
ls gpurun_out | head -40
