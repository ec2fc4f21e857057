#!/bin/bash
# dev: eager step time with / without the direct (dispatcher-free) calls of the cores
for d in 0 1 0 1; do EA_DIRECT_IMPL=$d python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('direct=$d', round(d['ms_per_step'],4), 'eager', d.get('eager_ms_per_step'), d.get('eager_ms_per_step_blocks'))"; done
