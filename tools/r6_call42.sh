#!/bin/bash
# round 6, GPU call 42: after the last small commits (per-thread hint, switch-aware tests): causal-EVA / primitive tests, smoke, LM line
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_causal_eva.py tests/test_gpu_primitives.py tests/test_gpu_modules.py -q -m gpu -n 2 > gpurun_out/gpu_tests42.log 2>&1; echo "rc $?" >> gpurun_out/gpu_tests42.log; tail -3 gpurun_out/gpu_tests42.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py --attn causal_eva --workload lm --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lm', d['ms_per_step'], d.get('ms_per_step_blocks'))"
