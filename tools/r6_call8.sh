#!/bin/bash
# round 6, GPU call 8: the round's evidence with the final library
bash tools/profiles_round.sh r06 > gpurun_out/profiles_round.log 2>&1
bash tools/profile_bench.sh causal_eva r06lm "--workload lm" > gpurun_out/prof_r06lm_causal_eva.log 2>&1
bash tools/pmc_sq.sh causal_eva lm > gpurun_out/sq_causal_eva_lm.log 2>&1
du -sh gpurun_out; ls gpurun_out | head -50
