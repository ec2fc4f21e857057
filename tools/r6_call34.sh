#!/bin/bash
# round 6, GPU call 34: the suite with the two-stage colsum view, the LM / cfg5 lines next to it
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu -n 2 > gpurun_out/gpu_tests34.log 2>&1; echo "rc $?" >> gpurun_out/gpu_tests34.log; tail -4 gpurun_out/gpu_tests34.log
for sw in 1 0; do
  EA_COLSUM_TWO_STAGE=$sw python bench.py --attn causal_eva --workload lm --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lm two_stage=$sw', d['ms_per_step'], d.get('ms_per_step_blocks'))"
  EA_COLSUM_TWO_STAGE=$sw python bench.py --attn eva --workload cfg5 --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 eva two_stage=$sw', d['ms_per_step'], d.get('ms_per_step_blocks'))"
done 2>&1 | tee gpurun_out/ab34.txt
