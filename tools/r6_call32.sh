#!/bin/bash
# round 6, GPU call 32: launch-ordered traces of one eager step -- cfg5 LARA at the recipe's batch of one, cfg5 LARA B = 16, cfg2 LARA
mkdir -p gpurun_out
bash tools/step_trace.sh gpurun_out/st32_cfg5_b1.txt --attn lara --workload cfg5 --batch 1
bash tools/step_trace.sh gpurun_out/st32_cfg5_b16.txt --attn lara --workload cfg5
bash tools/step_trace.sh gpurun_out/st32_cfg2.txt --attn lara --workload cfg2
tail -3 gpurun_out/st32_cfg5_b1.txt
