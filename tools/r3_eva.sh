#!/bin/bash
# dev: EVA / LARA parity + bench lines with a dev switch off / on  (usage: r3_eva.sh ENVVAR)
set -u
V=${1:-EA_LM_REG}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_modules.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py tests/test_gpu_padding.py tests/test_gpu_primitives.py -x -q -m gpu -k "eva or lara or chunk or beta or landmark" 2>&1 | tail -3
for ho in 0 1; do
for a in eva lara; do
env $V=$ho python bench.py --attn $a --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads > gpurun_out/r3_$a$ho.json 2> gpurun_out/r3_$a$ho.err; python - $ho $a $V <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r3_%s%s.json"%(sys.argv[2],sys.argv[1])).read().strip().splitlines()[-1])
print(sys.argv[3], sys.argv[1], sys.argv[2], "ms/step", round(d["ms_per_step"],4), "eager", d.get("eager_ms_per_step"))
k=d["roofline"]["all_kernels_avg_us"]; print({a:k[a] for a in k if "chunk" in a or "beta" in a or "window" in a})
PY
done
done
