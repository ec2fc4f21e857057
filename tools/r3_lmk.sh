#!/bin/bash
# dev: landmark-kernel parity + A/B bench lines against tools/bin/libea_hip_base.so
set -u
python -m pytest tests/test_gpu_modules.py tests/test_gpu_fullsize.py tests/test_gpu_custom_ops.py tests/test_gpu_configs.py tests/test_gpu_primitives.py tests/test_gpu_padding.py -x -q -m gpu -k "lara or eva or scatter or landmark" 2>&1 | tail -3
bash tools/r3_ab.sh lara eva
for lib in tools/bin/libea_hip_base.so efficient-attention_amd/lib/libea_hip.so; do
EA_HIP_LIB=$PWD/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_kernels_avg_us']; print('$lib'.split('/')[-1], {a:k[a] for a in k if 'landmarks' in a})"
done
