#!/bin/bash
# round 6, GPU call 38: final checks after the Python-side changes of the last session (suite, smoke, default bench)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu -n 2 > gpurun_out/gpu_tests38.log 2>&1; echo "rc $?" >> gpurun_out/gpu_tests38.log; tail -4 gpurun_out/gpu_tests38.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke38.log 2>&1; tail -1 gpurun_out/smoke38.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench38.json 2> gpurun_out/bench38.err; tail -c 300 gpurun_out/bench38.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench38.json').read().strip().splitlines()[-1])
print(d['metric'], d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['roofline'].get('avg_us'), d['roofline'].get('traffic'))
for k,v in (d.get('other_workloads') or {}).items():
    print(k, v.get('ms_per_step'), round(v.get('tokens_per_s',0)/1e6,2))
PY
