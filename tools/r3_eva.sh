#!/bin/bash
# dev: EVA / local parity + bench lines with and without the P / dS hand-over
set -u
mkdir -p gpurun_out
python -m pytest tests/test_gpu_modules.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py tests/test_gpu_window_sweep.py -x -q -m gpu -k "eva or local" 2>&1 | tail -3
for ho in 0 1; do
EA_WIN_HANDOVER=$ho python bench.py --attn eva --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads > gpurun_out/r3_eva$ho.json 2> gpurun_out/r3_eva$ho.err; python - $ho <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r3_eva%s.json"%sys.argv[1]).read().strip().splitlines()[-1])
print("handover", sys.argv[1], "EVA ms/step", round(d["ms_per_step"],4), "eager", d.get("eager_ms_per_step"))
k=d["roofline"]["all_kernels_avg_us"]; print({a:k[a] for a in k if "window" in a})
PY
done
