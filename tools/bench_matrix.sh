#!/bin/bash
# dev: the workload x attention table of DESIGN.md 5 (run on the GPU box)
for w in cfg3 cfg2 cfg5; do
  for a in lara eva softmax local performer ra scatterbrain; do
    python bench.py --attn $a --workload $w --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', '$a', round(d['value']/1e6,1), round(d['ms_per_step'],3), 'eager', d.get('eager_ms_per_step'))"
  done
done
python bench.py --attn causal_eva --workload lm --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lm causal_eva', round(d['value']/1e6,2), round(d['ms_per_step'],3))"
