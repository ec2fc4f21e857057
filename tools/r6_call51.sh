#!/bin/bash
# round 6, GPU call 51: launch-ordered traces of one eager step of cfg5 EVA and cfg3 EVA
mkdir -p gpurun_out
bash tools/step_trace.sh gpurun_out/st51_cfg5_eva.txt --attn eva --workload cfg5
bash tools/step_trace.sh gpurun_out/st51_cfg3_eva.txt --attn eva
tail -2 gpurun_out/st51_cfg5_eva.txt
