"""dev tool (GPU): time ea_wgrad against the library split-K path at the bench shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-attention_amd")]
import torch
from efficient_attention import _ops
shapes = [(100352, 576, 192), (100352, 192, 192)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for rows, M, K in shapes:
    dy = torch.randn(rows, M, device="cuda").bfloat16()
    x = torch.randn(rows, K, device="cuda").bfloat16()
    for name, fn in (("ea_wgrad", lambda: _ops.wgrad(dy, x, True)),
                     ("ea_wgrad_nobias", lambda: _ops.wgrad(dy, x, False)),
                     ("torch dy^T x", lambda: dy.t() @ x)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(rows, M, K, name, "%.1f us" % (e0.elapsed_time(e1) / 20 * 1e3), "S =", _ops.nv.lib().ea_wgrad_parts(rows, M, K))
