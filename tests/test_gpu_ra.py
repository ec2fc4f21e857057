"""-m gpu: randomized attention beyond the fixtures -- the Gumbel-max draw (ea_softmax_sample) follows
softmax(s q k^T), is reproducible from torch's generator, and the module runs end to end with its
own draws; a larger geometry against the oracle with shared draws."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-attention_amd"), os.path.join(ROOT, "tests")]


@pytest.mark.gpu
def test_sample_follows_softmax():
    from efficient_attention import _ops
    B, h, N, d = 1, 2, 40, 64
    gen = torch.Generator(device="cuda").manual_seed(0)
    q = (1.5 * torch.randn(B, h, N, d, device="cuda", generator=gen)).to(torch.bfloat16)
    k = (1.5 * torch.randn(B, h, N, d, device="cuda", generator=gen)).to(torch.bfloat16)
    pi = torch.softmax(d ** -0.5 * q.float() @ k.float().transpose(-1, -2), -1)      # [B,h,N,N]
    draws = 4096
    counts = torch.zeros(B, h, N, N, device="cuda")
    torch.manual_seed(123)
    for _ in range(draws):
        idx = _ops.softmax_sample(q, k)
        assert idx.shape == (B, h, N) and idx.dtype == torch.int64 and int(idx.min()) >= 0 and int(idx.max()) < N
        counts.scatter_add_(-1, idx.unsqueeze(-1), torch.ones(B, h, N, 1, device="cuda"))
    emp = counts / draws
    tv = 0.5 * (emp - pi).abs().sum(-1)                   # total variation per query
    # 40 outcomes, 4096 draws: E[TV] ~ 0.03 for a flat distribution, less for a peaked one
    assert tv.max().item() < 0.08, tv.max().item()
    # every probability mass above 1 % is hit within 5 sigma
    sigma = (pi * (1 - pi) / draws).sqrt()
    assert ((emp - pi).abs() <= 5 * sigma + 1e-3).all()
    # same generator state -> same draws
    torch.manual_seed(7)
    a = _ops.softmax_sample(q, k)
    torch.manual_seed(7)
    b = _ops.softmax_sample(q, k)
    assert torch.equal(a, b)
    assert not torch.equal(a, _ops.softmax_sample(q, k))


@pytest.mark.gpu
@pytest.mark.parametrize("num_samples", [1, -1, 0])
def test_module_matches_oracle_at_784(num_samples):
    import efficient_attention as ea
    import oracle
    from gpu_checks import MODULE_TOL
    from util import scaled_err
    torch.manual_seed(5)
    args = dict(dim=192, num_heads=3, num_samples=num_samples)
    m = ea.AttentionFactory.build_attention("ra", dict(args)).cuda().eval()
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.05 * torch.randn_like(p))
    B = 2
    x = torch.randn(B, 28, 28, 192, device="cuda").requires_grad_(True)
    g = torch.randn(B, 28, 28, 192, device="cuda")
    draws = torch.randint(0, 784, (B, 3, 784), device="cuda")
    m._sample_index_fn = lambda shape: draws
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = m(x)
    (y.float() * g).sum().backward()
    params = {k: v.detach().float().cpu().requires_grad_(True) for k, v in m.state_dict().items()}
    xr = x.detach().cpu().requires_grad_(True)
    yr = oracle.module_forward("ra", args, params, xr, None, training=False, index_fn=lambda shape: draws.cpu())
    (yr * g.cpu()).sum().backward()
    errs = {"y": scaled_err(y.detach().float().cpu().numpy(), yr.detach().numpy()),
            "dx": scaled_err(x.grad.cpu().numpy(), xr.grad.numpy())}
    for k, p in m.named_parameters():
        errs["d" + k] = scaled_err(p.grad.float().cpu().numpy(), params[k].grad.numpy())
    from gpu_checks import tol_for
    tol = tol_for("ra", "bf16", "test_gpu_ra")
    bad = {k: v for k, v in errs.items() if not (v[0] <= tol[0] and v[1] <= tol[1])}
    assert not bad, (num_samples, bad)


@pytest.mark.gpu
def test_module_runs_with_its_own_draws():
    import efficient_attention as ea
    m = ea.AttentionFactory.build_attention("ra", dict(dim=128, num_heads=2, num_samples=1)).cuda().train()
    x = torch.randn(3, 100, 128, device="cuda", requires_grad=True)
    torch.manual_seed(11)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y1 = m(x)
    y1.float().square().mean().backward()
    assert torch.isfinite(y1).all() and torch.isfinite(x.grad).all()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    torch.manual_seed(11)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y2 = m(x)
    assert torch.equal(y1, y2)
