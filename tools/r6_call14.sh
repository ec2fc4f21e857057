#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { # label, env, bench args
  env $2 python bench.py $3 --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$3', round(d['ms_per_step'],4), (d.get('ms_per_step_blocks') or {}).get('median'), {k: v for k, v in d['roofline']['all_kernels_avg_us'].items() if 'k_fused' in k})"
}
for rep in 1 2; do
  for spec in "--attn lara" "--attn lara --workload cfg2"; do
    for v in 0 64 128 192 256; do
      run x$v "EA_LARA_FK_X=$v" "$spec"
    done
  done
done > gpurun_out/ab14.log 2>&1
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_modules.py tests/test_gpu_configs.py tests/test_gpu_window_sweep.py -q -m gpu -x > gpurun_out/gpu_tests14.log 2>&1; echo "pytest rc $?" >> gpurun_out/gpu_tests14.log
cat gpurun_out/ab14.log | cut -c1-200; tail -3 gpurun_out/gpu_tests14.log
