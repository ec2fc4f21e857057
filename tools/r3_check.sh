#!/bin/bash
# round-3 dev loop on the GPU box: targeted parity tests, then bench lines, optional SQ counters / stamps
set -u
mkdir -p gpurun_out
python -m pytest tests/test_gpu_primitives.py -x -q -m gpu -k "linear or colsum or wgrad" 2>&1 | tail -3
python -m pytest tests/test_gpu_modules.py tests/test_gpu_fullsize.py tests/test_gpu_custom_ops.py tests/test_gpu_configs.py -x -q -m gpu -k "lara" 2>&1 | tail -3
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r3_lara.json 2> gpurun_out/r3_lara.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3_lara.json").read().strip().splitlines()[-1])
print("LARA ms/step", round(d["ms_per_step"],4), "eager", d.get("eager_ms_per_step"))
print({k:v for k,v in d["roofline"]["all_kernels_avg_us"].items()})
PY
if [ "${1:-}" = "sq" ]; then
  bash tools/pmc_sq.sh lara > /dev/null 2>&1
  EA_HIP_LIB=$PWD/tools/bin/libea_hip_prof.so python tools/time_lara.py lara 2 2> gpurun_out/stamps_lara.txt >/dev/null
  grep -A1 "lara_f mode" gpurun_out/stamps_lara.txt | tail -6 | cut -c1-700
fi
