#!/bin/bash
# round 6, GPU call 25: the round's evidence with the FINAL library (kernel trace, HBM and SQ counters for LARA / EVA at
# N = 196 / 784 / 4096, softmax at 784 / 4096, the LM layer)
export TMPDIR=/tmp
mkdir -p gpurun_out
sha256sum efficient-attention_amd/lib/libea_hip.so > gpurun_out/final_lib_sha.txt
bash tools/profiles_round.sh r06 > gpurun_out/profiles_round.log 2>&1
bash tools/profile_bench.sh causal_eva r06lm "--workload lm" > gpurun_out/prof_r06lm_causal_eva.log 2>&1
bash tools/pmc_sq.sh causal_eva lm > gpurun_out/sq_causal_eva_lm.log 2>&1
du -sh gpurun_out; ls gpurun_out | head -70
