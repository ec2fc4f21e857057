#!/bin/bash
# round 6, GPU call 36: launch-ordered trace of one eager LM step
mkdir -p gpurun_out
bash tools/step_trace.sh gpurun_out/st36_lm.txt --attn causal_eva --workload lm
tail -3 gpurun_out/st36_lm.txt
