"""dev tool (GPU): ScatterBrain feature half, HIP kernels vs the torch-op path on the same qkv."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'efficient-attention_amd'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import numpy as np, torch, warnings
warnings.simplefilter("ignore")
import efficient_attention as ea
from efficient_attention import _ops
from util import scaled_err
torch.manual_seed(0)
dt = torch.float16 if len(sys.argv) > 1 and sys.argv[1] == "fp16" else torch.bfloat16
m = ea.AttentionFactory.build_attention("scatterbrain", dict(dim=128, num_heads=2, window_size=7, attn_2d=True, use_rpe=True, approx_attn_dim=64)).cuda().eval()
B, N, h, d = 2, 196, 2, 64
qkv = (0.7 * torch.randn(B, N, 3, h, d, device="cuda")).to(dt)
g = torch.randn(B, N, h, d, device="cuda").to(dt)
res = {}
for mode in ("hip", "torch"):
    _ops.SCATTER_TORCH = mode == "torch"
    x = qkv.clone().requires_grad_(True)
    out = m._scatter(x, None, [14, 14])
    (out.float() * g.float()).sum().backward()
    res[mode] = (out.detach().float(), x.grad.float())
print("out", scaled_err(res["hip"][0].cpu().numpy(), res["torch"][0].cpu().numpy()))
for i, nm in enumerate("qkv"):
    a, b = res["hip"][1][:, :, i], res["torch"][1][:, :, i]
    print("d" + nm, scaled_err(a.cpu().numpy(), b.cpu().numpy()), float(a.abs().max()), float(b.abs().max()))
