#!/bin/bash
# dev: per-kernel event timings (bench.py's instrumented pass) for several library builds on one box
#   r4_kern.sh "lib1.so lib2.so" "<bench args>" ...
LIBS=$1; shift
for spec in "$@"; do
for lib in $LIBS; do
EA_HIP_LIB=$PWD/$lib python bench.py $spec --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$spec', '$lib'.split('/')[-1], round(d['ms_per_step'],4), d['eager_ms_per_step'], d['roofline']['all_kernels_avg_us'])"
done; done
