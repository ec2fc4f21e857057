#!/bin/bash
# round 6, GPU call 28: launch-ordered traces of the PvT stage 3 / 4 layers (cfg4) and the cfg2 layers
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/step_trace.sh gpurun_out/trace28_s3.txt --attn eva --batch 32 --dim 320 --heads 5 --grid 24 --window 8 --landmarks 36
bash tools/step_trace.sh gpurun_out/trace28_s4.txt --attn softmax --batch 32 --dim 512 --heads 8 --grid 12
bash tools/step_trace.sh gpurun_out/trace28_cfg2_eva.txt --attn eva --workload cfg2
bash tools/step_trace.sh gpurun_out/trace28_cfg2_lara.txt --attn lara --workload cfg2
tail -n 2 gpurun_out/trace28_s3.txt
