#!/bin/bash
# round 6, GPU call 53: the last checks of the round -- the driver's sequential suite command, smoke, the default bench line
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1150 python -m pytest tests/ -x -q -m gpu > gpurun_out/gpu_tests53.log 2>&1; echo "rc $?" >> gpurun_out/gpu_tests53.log; grep -E "^FAILED|passed|failed|rc " gpurun_out/gpu_tests53.log | tail -5
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke53.log 2>&1; tail -1 gpurun_out/smoke53.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench53.json 2> gpurun_out/bench53.err; tail -c 200 gpurun_out/bench53.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench53.json').read().strip().splitlines()[-1])
print(d['metric'], d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['roofline'].get('frac_rocprof'), d['roofline'].get('traffic'))
for k,v in (d.get('other_workloads') or {}).items():
    print(k, v.get('ms_per_step'), round(v.get('tokens_per_s',0)/1e6,2), v.get('error'))
PY
bash tools/profile_bench.sh eva r06cfg5 "--workload cfg5" > gpurun_out/prof_r06cfg5_eva.log 2>&1
