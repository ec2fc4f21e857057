for rep in 1 2; do
for fin in 1 0; do
for attn in lara eva; do
EA_DGRAD_FIN=$fin python bench.py --attn $attn --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_kernels_avg_us']
print('$attn fin=$fin', round(d['ms_per_step'],4), (d.get('ms_per_step_blocks') or {}).get('median'), {n: v for n, v in k.items() if 'dgrad' in n or 'finish' in n or 'chunk_mean_bwd' in n})"
done; done; done
