import sys, os, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-attention_amd")]
import torch
import efficient_attention as ea
warnings.simplefilter("ignore")
torch.manual_seed(0)
x = torch.randn(2, 24, 128, device="cuda")
mask = torch.zeros(2, 24, dtype=torch.bool, device="cuda"); mask[1, :] = True
for attn, args in (("softmax", dict(dim=128, num_heads=2)),
                   ("lara", dict(dim=128, num_heads=2, num_landmarks=4, proposal_gen="adaptive-1d", mis_type="mis-opt")),
                   ("ra", dict(dim=128, num_heads=2)),
                   ("scatterbrain", dict(dim=128, num_heads=2, window_size=8, attn_2d=False, approx_attn_dim=16))):
    for train in (False, True):
        try:
            m = ea.AttentionFactory.build_attention(attn, dict(args)).cuda().train(train)
            xx = x.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m(xx, mask)
            y[0].float().sum().backward()
            print(attn, "train" if train else "eval", "row0 finite", torch.isfinite(y[0]).all().item(), "row1 nan-all", torch.isnan(y[1]).all().item(),
                  "row1 finite", torch.isfinite(y[1]).all().item(), "dx0 finite", torch.isfinite(xx.grad[0]).all().item(), "dx1 finite", torch.isfinite(xx.grad[1]).all().item(),
                  "pgrad finite", all(torch.isfinite(p.grad).all().item() for p in m.parameters() if p.grad is not None))
        except Exception as e:
            print(attn, "ERR", repr(e)[:300])
