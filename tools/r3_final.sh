#!/bin/bash
# end-of-round check on the GPU box: smoke, full GPU suite, the driver's default bench line
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python bench.py > gpurun_out/r3_default.json 2> gpurun_out/r3_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3_default.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("metric","value","unit","n_gpus","steps","warmup","ms_per_step","dtype","vs_baseline","scaling")})
print("roofline", {k:v for k,v in d["roofline"].items() if k!="all_kernels_avg_us"})
print("cpu_baseline", d["cpu_baseline"])
print("eager", d.get("eager_ms_per_step"), "blocks", d.get("ms_per_step_blocks"))
for k,v in (d.get("other_workloads") or {}).items(): print(k, round(v["tokens_per_s"]/1e6,1), v["ms_per_step"], v["layer_hbm_roofline_frac"])
PY
