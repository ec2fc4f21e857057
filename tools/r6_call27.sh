#!/bin/bash
# round 6, GPU call 27: side stream for the landmark noise (next to the projection) and the output projection's weight gradient
# (next to the estimator's backward): A/B on one box
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { # label env args
  env $2 python bench.py $3 --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$3', round(d['ms_per_step'],4), (d.get('ms_per_step_blocks') or {}).get('median'))"
}
for rep in 1 2; do
  for spec in "--attn lara" "--attn lara --workload cfg2"; do
    run both "EA_SIDE_STREAM=1" "$spec"
    run noise_only "EA_SIDE_WGRAD=0" "$spec"
    run off "EA_SIDE_STREAM=0" "$spec"
  done
done > gpurun_out/ab27.log 2>&1
cat gpurun_out/ab27.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_modules.py tests/test_gpu_configs.py tests/test_gpu_harness.py -q -m gpu -x -k "lara" > gpurun_out/t27.log 2>&1; echo "rc $?" >> gpurun_out/t27.log; tail -3 gpurun_out/t27.log
