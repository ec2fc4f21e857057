#!/bin/bash
# dev: EVA / local parity + bench lines with a dev switch off / on  (usage: r3_eva.sh ENVVAR)
set -u
V=${1:-EA_WIN_PLAIN}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_modules.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py tests/test_gpu_window_sweep.py tests/test_gpu_padding.py -x -q -m gpu -k "eva or local or scatter" 2>&1 | tail -3
for rep in 1 2; do
for ho in 0 1; do
for a in eva local; do
env $V=$ho python bench.py --attn $a --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads > gpurun_out/r3_$a$ho.json 2> gpurun_out/r3_$a$ho.err; python - $ho $a $V <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r3_%s%s.json"%(sys.argv[2],sys.argv[1])).read().strip().splitlines()[-1])
k=d["roofline"]["all_kernels_avg_us"]
print(sys.argv[3], sys.argv[1], sys.argv[2], "ms/step", round(d["ms_per_step"],4), {a:k[a] for a in k if "window" in a})
PY
done
done
done
