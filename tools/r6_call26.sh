#!/bin/bash
# round 6, GPU call 26: final checks -- smoke, the default bench run (what the driver runs), the suite under EA_SEGLIN_FIN=0
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke26.log 2>&1; tail -2 gpurun_out/smoke26.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench26.json 2> gpurun_out/bench26.err; tail -c 600 gpurun_out/bench26.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench26.json').read().strip().splitlines()[-1])
print(d['metric'], d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['roofline'].get('avg_us'))
for k,v in (d.get('other_workloads') or {}).items():
    print(k, v.get('value'), v.get('ms_per_step'))
PY
EA_SEGLIN_FIN=0 timeout 1500 python -m pytest tests -q -m gpu -n 2 > gpurun_out/gpu_tests26_fin0.log 2>&1; echo "rc $?" >> gpurun_out/gpu_tests26_fin0.log; tail -3 gpurun_out/gpu_tests26_fin0.log
