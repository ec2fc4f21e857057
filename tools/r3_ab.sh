#!/bin/bash
# dev: A/B of two builds of the library in one box (tools/bin/libea_hip_base.so vs the in-tree one)
#   usage: r3_ab.sh "<attn> [bench flags]" ...
for spec in "$@"; do
for rep in 1 2; do
for lib in tools/bin/libea_hip_base.so efficient-attention_amd/lib/libea_hip.so; do
EA_HIP_LIB=$PWD/$lib python bench.py --attn $spec --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$spec', '$lib'.split('/')[-1], round(d['ms_per_step'],4), (d.get('ms_per_step_blocks') or {}).get('median'))"
done; done; done
