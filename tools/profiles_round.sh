#!/bin/bash
# all committed evidence of a round in one GPU call: kernel-trace stats + HBM PMC (profile_bench.sh) and SQ/MFMA counters (pmc_sq.sh)
# for LARA and EVA at north_star's three sizes (N = 196: cfg2, 784: cfg3 = the default, 4096: cfg5) and the softmax baseline
# at N = 784 / 4096.  usage: tools/profiles_round.sh <tag> ; afterwards, per (attn, size):
#   python tools/summarize_profile.py gpurun_out/prof_<tag>[cfgN]_<attn> <tag>[cfgN] <attn> profiles [cfg2|cfg5]
TAG=${1:-r06}
for a in lara eva; do
  bash tools/profile_bench.sh $a $TAG > gpurun_out/prof_${TAG}_$a.log 2>&1
  for wl in cfg2 cfg5; do
    bash tools/profile_bench.sh $a ${TAG}$wl "--workload $wl" > gpurun_out/prof_${TAG}${wl}_$a.log 2>&1
  done
done
bash tools/profile_bench.sh softmax $TAG "--no-other-workloads" > gpurun_out/prof_${TAG}_softmax.log 2>&1
bash tools/profile_bench.sh softmax ${TAG}cfg5 "--workload cfg5" > gpurun_out/prof_${TAG}cfg5_softmax.log 2>&1
for a in lara eva softmax; do bash tools/pmc_sq.sh $a > gpurun_out/sq_$a.log 2>&1; done
bash tools/pmc_sq.sh softmax cfg5 > gpurun_out/sq_softmax_cfg5.log 2>&1
ls gpurun_out | head -60
