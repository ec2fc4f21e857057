import sys, os, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-attention_amd")]
import torch
import efficient_attention as ea
warnings.simplefilter("ignore")
torch.manual_seed(0)
m = ea.AttentionFactory.build_attention("performer", dict(dim=128, num_heads=2, approx_attn_dim=64, proj_method="favorp")).cuda().eval()
x = (0.25 * torch.randn(1, 14, 14, 128, device="cuda")).requires_grad_(True)
qkv_grads = {}
def hook(mod, gi, go): pass
with torch.autocast("cuda", dtype=torch.bfloat16):
    y = m(x)
y.float().sum().backward()
g = m.qkv.bias.grad.view(3, -1)
print("y finite", torch.isfinite(y).all().item())
for i, n in enumerate("qkv"):
    print("d%s bias finite" % n, torch.isfinite(g[i]).all().item(), g[i][:4].tolist())
from efficient_attention import _ops
qkv5 = (0.25 * torch.randn(1, 196, 3, 2, 64, device="cuda")).bfloat16()
W = torch.randn(2, 64, 64, device="cuda")
out, stab, kv, ksum = torch.ops.ea.performer_fwd(qkv5, None, W)
dout = torch.randn_like(out)
dqkv = torch.ops.ea.performer_bwd(dout, qkv5, None, W, stab, kv, ksum, out)
dq = dqkv[0, :, 0]   # [N, h, d]
bad = ~torch.isfinite(dq.float())
print("bad count", bad.sum().item(), "of", bad.numel())
idx = bad.nonzero()
print("tokens", sorted(set(idx[:, 0].tolist()))[:40])
print("heads", sorted(set(idx[:, 1].tolist())), "channels", sorted(set(idx[:, 2].tolist()))[:70])
