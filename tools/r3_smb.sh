python -m pytest tests/test_gpu_modules.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py tests/test_gpu_ra.py tests/test_gpu_properties.py tests/test_gpu_padding.py -x -q -m gpu -k "softmax or ra or sample" 2>&1 | tail -3
for qt in 1 2; do for w in cfg5 cfg3; do EA_SM_BQT=$qt python bench.py --attn softmax --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_kernels_avg_us']
print('bqt=$qt $w', round(d['ms_per_step'],4), round(d['value']/1e6,1), {n:k[n] for n in k if 'softmax' in n})"; done; done
