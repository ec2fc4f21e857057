"""Per-kernel timing of the LARA module at the bench shape (dev tool, GPU only): runs fwd+bwd N times."""
import sys, os, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-attention_amd")]
import torch
import efficient_attention as ea
from efficient_attention import _ops
warnings.simplefilter("ignore")
attn = sys.argv[1] if len(sys.argv) > 1 else "lara"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
import bench
wl = sys.argv[3] if len(sys.argv) > 3 else "cfg3"            # cfg3 (default) | cfg5 | lm (with causal_eva)
call = lambda m, x: m(x)
if wl == "lm":
    m = bench.build_layer(attn, 1024, 8, (512,), "cuda"); m.train()
    x = torch.randn(512, 18, 1024, device="cuda", requires_grad=True)
    g = torch.randn(512, 18, 1024, device="cuda").bfloat16()
    call = lambda m, x: m(x, x, x)[0]
elif wl == "cfg5":
    m = bench.build_layer(attn, 512, 8, (4096,), "cuda"); m.train()
    x = torch.randn(16, 4096, 512, device="cuda", requires_grad=True)
    g = torch.randn(16, 4096, 512, device="cuda").bfloat16()
else:
    m = bench.build_layer(attn, 192, 3, 28, "cuda"); m.train()
    x = torch.randn(128, 28, 28, 192, device="cuda", requires_grad=True)
    g = torch.randn(128, 28, 28, 192, device="cuda").bfloat16()
for i in range(n):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = call(m, x)
    y.backward(g)
torch.cuda.synchronize()
print("done")
