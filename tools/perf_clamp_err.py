import sys, os
sys.path[:0]=['/root/repo','/root/repo/efficient-attention_amd','/root/repo/tests','/root/repo/tests/golden']
import torch
from gpu_checks import check_module_case
for dt in (torch.bfloat16, torch.float16):
    for mode in ("eval","train"):
        e=check_module_case("performer_2d_clamp", mode, dtype=dt, tol=(1.0,1.0))
        print(dt, mode, {k:(round(float(v[0]),4), round(float(v[1]),4)) for k,v in e.items()})
