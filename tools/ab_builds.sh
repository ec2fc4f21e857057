#!/bin/bash
# dev: A/B/C... of several builds of the library on ONE box
#   usage: r4_ab.sh "lib1.so lib2.so ..." "<attn> [bench flags]" ...
LIBS=$1; shift
for spec in "$@"; do
for rep in 1 2; do
for lib in $LIBS; do
EA_HIP_LIB=$PWD/$lib python bench.py --attn $spec --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$spec', '$lib'.split('/')[-1], round(d['ms_per_step'],4), (d.get('ms_per_step_blocks') or {}).get('median'), {k: v for k, v in d['roofline']['all_kernels_avg_us'].items()} if '$rep' == '2' else '')"
done; done; done
