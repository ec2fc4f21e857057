#!/bin/bash
# round 6, GPU call 21: launch-ordered traces of the LM layer, cfg5 LARA at batch one and cfg5 EVA
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/step_trace.sh gpurun_out/trace21_lm.txt --attn causal_eva --workload lm
bash tools/step_trace.sh gpurun_out/trace21_cfg5_b1.txt --attn lara --workload cfg5 --batch 1
bash tools/step_trace.sh gpurun_out/trace21_cfg5_eva.txt --attn eva --workload cfg5
tail -3 gpurun_out/trace21_*.txt | cut -c1-150
