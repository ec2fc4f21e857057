#!/bin/bash
# dev: landmark-kernel parity + stamps + bench lines
set -u
mkdir -p gpurun_out
python -m pytest tests/test_gpu_primitives.py -x -q -m gpu -k "landmark or lmk" 2>&1 | tail -3
python -m pytest tests/test_gpu_modules.py tests/test_gpu_fullsize.py tests/test_gpu_custom_ops.py tests/test_gpu_configs.py tests/test_gpu_scatter.py -x -q -m gpu -k "lara or eva or scatter" 2>&1 | tail -3
EA_HIP_LIB=$PWD/tools/bin/libea_hip_prof.so python tools/time_lara.py lara 2 2> gpurun_out/stamps_lmk.txt >/dev/null
grep -i "lmk" gpurun_out/stamps_lmk.txt | tail -2 | cut -c1-800
for a in lara eva; do
python bench.py --attn $a --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads > gpurun_out/r3_$a.json 2> gpurun_out/r3_$a.err; python - $a <<'PY'
import json,sys
a=sys.argv[1]
d=json.loads(open("gpurun_out/r3_%s.json"%a).read().strip().splitlines()[-1])
print(a, "ms/step", round(d["ms_per_step"],4), "eager", d.get("eager_ms_per_step"))
print({k:v for k,v in d["roofline"]["all_kernels_avg_us"].items()})
PY
done
