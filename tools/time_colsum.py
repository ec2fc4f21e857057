import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-attention_amd")]
import torch
from efficient_attention import _ops
def tm(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    big = torch.empty(600*1024*1024, dtype=torch.uint8, device="cuda")
    tot=0
    for _ in range(n):
        big.fill_(1)   # flush MALL/L2
        s.record(); fn(); e.record(); torch.cuda.synchronize(); tot += s.elapsed_time(e)
    return tot/n*1e3
for cols in (576, 192):
    dy = torch.randn(100352, cols, device="cuda").bfloat16()
    print(cols, "ea %.1f us   torch %.1f us" % (tm(lambda: _ops.bias_grad(dy)), tm(lambda: dy.sum(0, dtype=torch.float32))))
