#!/bin/bash
# round 6, GPU call 48: final checks with the single-node paths of the last session (suite, smoke, default bench), then the evidence of
# the two workloads whose launch structure changed (cfg5 LARA, the LM layer) regenerated
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -n 2 > gpurun_out/gpu_tests48.log 2>&1; echo "rc $?" >> gpurun_out/gpu_tests48.log; grep -E "^FAILED|passed|failed" gpurun_out/gpu_tests48.log | tail -6
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke48.log 2>&1; tail -1 gpurun_out/smoke48.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench48.json 2> gpurun_out/bench48.err; tail -c 300 gpurun_out/bench48.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench48.json').read().strip().splitlines()[-1])
print(d['metric'], d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['roofline'].get('avg_us'), d['roofline'].get('traffic'))
for k,v in (d.get('other_workloads') or {}).items():
    print(k, v.get('ms_per_step'), round(v.get('tokens_per_s',0)/1e6,2), v.get('error'))
PY
bash tools/profile_bench.sh lara r06cfg5 "--workload cfg5" > gpurun_out/prof_r06cfg5_lara.log 2>&1
bash tools/profile_bench.sh causal_eva r06lm "--workload lm" > gpurun_out/prof_r06lm_causal_eva.log 2>&1
ls gpurun_out | head -30
