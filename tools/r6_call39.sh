#!/bin/bash
# round 6, GPU call 39: the LM layer's evidence regenerated after the Python-side changes of its step (kernel trace + HBM counters)
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/profile_bench.sh causal_eva r06lm "--workload lm" > gpurun_out/prof_r06lm_causal_eva.log 2>&1
tail -5 gpurun_out/prof_r06lm_causal_eva.log
