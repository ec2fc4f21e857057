#!/bin/bash
mkdir -p gpurun_out
bash tools/step_trace.sh gpurun_out/st_lara_cfg5_b1.txt --attn lara --workload cfg5 --batch 1
bash tools/step_trace.sh gpurun_out/st_lara_cfg5_b16.txt --attn lara --workload cfg5
bash tools/step_trace.sh gpurun_out/st_lm.txt --attn causal_eva --workload lm
bash tools/step_trace.sh gpurun_out/st_eva_s3.txt --attn eva --batch 32 --grid 24 --dim 320 --heads 5 --window 8 --landmarks 36
bash tools/step_trace.sh gpurun_out/st_eva_cfg5.txt --attn eva --workload cfg5
bash tools/step_trace.sh gpurun_out/st_softmax_s4.txt --attn softmax --batch 32 --grid 12 --dim 512 --heads 8
for f in gpurun_out/st_*.txt; do tail -n 2 $f; done; tail -n 5 gpurun_out/st_lm.txt.log
