#!/bin/bash
# round 6, GPU call 31: the suite with the new causal-EVA cases (quantization noise golden, padded decoding), smoke, default bench
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu -n 2 > gpurun_out/gpu_tests31.log 2>&1; echo "rc $?" >> gpurun_out/gpu_tests31.log; tail -15 gpurun_out/gpu_tests31.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke31.log 2>&1; tail -2 gpurun_out/smoke31.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench31.json 2> gpurun_out/bench31.err; tail -c 600 gpurun_out/bench31.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench31.json').read().strip().splitlines()[-1])
print(d['metric'], d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['roofline'].get('avg_us'), d['roofline'].get('traffic'))
for k,v in (d.get('other_workloads') or {}).items():
    print(k, v.get('value'), v.get('ms_per_step'))
PY
