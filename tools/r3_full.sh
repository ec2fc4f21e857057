#!/bin/bash
# full GPU suite + bench lines (round-3 dev loop)
set -u
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -6
for a in lara eva; do
python bench.py --attn $a --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r3_$a.json 2> gpurun_out/r3_$a.err; python - $a <<'PY'
import json,sys
a=sys.argv[1]
d=json.loads(open("gpurun_out/r3_%s.json"%a).read().strip().splitlines()[-1])
print(a, "ms/step", round(d["ms_per_step"],4), "eager", d.get("eager_ms_per_step"))
print({k:v for k,v in d["roofline"]["all_kernels_avg_us"].items()})
PY
done
