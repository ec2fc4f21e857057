#!/bin/bash
# round 6, GPU call 33: ea_colsum_f32 at the LM / cfg5 shapes, one stage against the folded two-stage view (tools/colsum_bench.py)
mkdir -p gpurun_out
python tools/colsum_bench.py > gpurun_out/colsum33.txt 2>&1; cat gpurun_out/colsum33.txt
