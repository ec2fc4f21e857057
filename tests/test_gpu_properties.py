"""-m gpu: size-independent properties of every variant at the BASELINE workload size
(x = [128, 28, 28, 192], h = 3, d = 64, bf16 autocast) -- sizes the CPU oracle cannot reach.
All of them go through the product nn.Module and therefore through the C ABI.

  * convexity : with v_n == c for every token the attention output is c (every estimator here is
                a normalised combination of values: softmax / window+control variates / LARA's
                self-normalised importance weights / Performer's ratio)
  * linearity : out is linear in the value projection, out(Wv1 + Wv2) == out(Wv1) + out(Wv2)
  * locality  : batch elements are independent, module(x)[b0:b1] == module(x[b0:b1]) (bitwise for
                the window / softmax kernels; to bf16 rounding where sequence slices are merged)
  * gradient of the convexity identity: with proj = I and loss = sum(y), d loss / d (v bias) == B*N
"""
import os
import sys
import warnings

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-attention_amd")]

B, G, C, H = 128, 28, 192, 3
VARIANTS = ["softmax", "local", "eva", "lara", "performer"]


def _layer(attn):
    import bench
    torch.manual_seed(11)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = bench.build_layer(attn, C, H, G, "cuda")
    m.eval()
    with torch.no_grad():
        for p in m.parameters():                    # make zero-initialised tables / biases matter
            p.add_(0.02 * torch.randn_like(p))
        m.proj.weight.copy_(torch.eye(C))
        m.proj.bias.zero_()
    return m


def _fwd(m, x):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        return m(x).float()


def _x(batch=B):
    g = torch.Generator(device="cuda").manual_seed(5)
    return torch.randn(batch, G, G, C, device="cuda", generator=g)


@pytest.mark.gpu
@pytest.mark.parametrize("attn", VARIANTS)
def test_constant_values_pass_through(attn):
    m = _layer(attn)
    cvec = torch.linspace(-1.0, 1.0, C, device="cuda")
    with torch.no_grad():
        m.qkv.weight[2 * C:].zero_()
        m.qkv.bias[2 * C:].copy_(cvec)
        y = _fwd(m, _x())
    ref = cvec.to(torch.bfloat16).float().expand_as(y)
    # bf16 probabilities / weights: each output is a convex combination of identical bf16 values
    assert torch.isfinite(y).all()
    assert (y - ref).abs().max().item() <= 2e-2, (y - ref).abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("attn", VARIANTS)
def test_linear_in_value_projection(attn):
    m = _layer(attn)
    x = _x()
    with torch.no_grad():
        wv = m.qkv.weight[2 * C:].clone()
        bv = m.qkv.bias[2 * C:].clone()
        g = torch.Generator(device="cuda").manual_seed(9)
        w2 = 0.05 * torch.randn(wv.shape, device="cuda", generator=g)
        y1 = _fwd(m, x)
        m.qkv.weight[2 * C:].copy_(w2); m.qkv.bias[2 * C:].zero_()
        y2 = _fwd(m, x)
        m.qkv.weight[2 * C:].copy_(wv + w2); m.qkv.bias[2 * C:].copy_(bv)
        y12 = _fwd(m, x)
    err = (y12 - (y1 + y2)).abs().max().item()
    scale = y12.abs().max().item()
    assert err <= 3e-2 * scale, (err, scale)


@pytest.mark.gpu
@pytest.mark.parametrize("attn", VARIANTS)
def test_batch_elements_are_independent(attn):
    m = _layer(attn)
    x = _x()
    with torch.no_grad():
        full = _fwd(m, x)
        part = _fwd(m, x[40:44].contiguous())
    err = (full[40:44] - part).abs().max().item()
    if attn in ("lara", "performer"):
        # sequence-wide sums are cut into a batch-size dependent number of slices whose partials are
        # merged in fp32 and rounded to bf16 once: results agree to bf16 rounding, not bitwise
        assert err <= 1.5e-2 * full.abs().max().item(), err
    else:
        assert err == 0.0, err


@pytest.mark.gpu
@pytest.mark.parametrize("attn", VARIANTS)
def test_value_bias_gradient_counts_tokens(attn):
    """y = c for v == c (convexity), so d sum(y) / d c_j = number of (batch, token) pairs: a
    whole-backward check (window / landmark / sequence-wide terms must cancel exactly in dq, dk)."""
    m = _layer(attn)
    bsz = 32
    with torch.no_grad():
        m.qkv.weight[2 * C:].zero_()
    x = _x(bsz)
    y = _fwd(m, x)
    y.sum().backward()
    gb = m.qkv.bias.grad[2 * C:].float()
    n = bsz * G * G
    assert (gb - n).abs().max().item() <= 2e-2 * n, ((gb - n).abs().max().item(), n)
    # q / k receive (numerically) no gradient: the output does not depend on the attention weights
    gq = m.qkv.bias.grad[:2 * C].float().abs().max().item()
    assert gq <= 2e-2 * n, gq


@pytest.mark.gpu
@pytest.mark.parametrize("attn", VARIANTS)
def test_full_batch_slice_matches_oracle(attn):
    """Forward and input gradient of two batch elements inside the full B = 128 launch (the launch
    geometry of the benchmark: two sequence slices per (b,h), uneven backward slices, one resident
    round of workgroups) against the CPU oracle run on just those two elements."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bench
    import oracle
    from gpu_checks import MODULE_TOL, LARA_TOL
    from util import scaled_err
    torch.manual_seed(11)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = bench.build_layer(attn, C, H, G, "cuda")
    m.eval()
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.02 * torch.randn_like(p))
    x = _x().requires_grad_(True)
    gen = torch.Generator(device="cuda").manual_seed(3)
    gy = torch.randn(B, G, G, C, device="cuda", generator=gen)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = m(x)
    (y.float() * gy).sum().backward()
    sl = slice(40, 42)
    params = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    xr = x.detach()[sl].cpu().requires_grad_(True)
    ref = oracle.module_forward(attn, bench.attn_args(attn, C, H, G), params, xr, None, training=False)
    (ref * gy[sl].cpu()).sum().backward()
    from gpu_checks import tol_for
    tol = tol_for(attn, "bf16", "test_gpu_properties")
    for name, got, want in (("y", y.detach().float()[sl].cpu(), ref.detach()), ("dx", x.grad[sl].cpu(), xr.grad)):
        e = scaled_err(got.numpy(), want.numpy())
        assert e[0] <= tol[0] and e[1] <= tol[1], (attn, name, e)


@pytest.mark.gpu
@pytest.mark.parametrize("attn", ["lara", "eva"])
def test_backward_is_loss_scale_invariant(attn):
    """The landmark pipeline multiplies fp16-rounded operands; its gradient-side matrices carry a
    per-matrix power-of-two scale so that a loss scale cannot push them out of fp16 range:
    backward(2^20 * g) == 2^20 * backward(g) (to rounding), and nothing overflows."""
    import bench
    torch.manual_seed(11)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = bench.build_layer(attn, C, H, G, "cuda")
    m.train()
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.02 * torch.randn_like(p))
    bsz = 16
    gen = torch.Generator(device="cuda").manual_seed(5)
    x0 = torch.randn(bsz, G, G, C, device="cuda", generator=gen)
    gy = torch.randn(bsz, G, G, C, device="cuda", generator=gen)
    grads = []
    for scale in (1.0, 2.0 ** 20):
        for p in m.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        torch.manual_seed(77)                       # same sampling noise in both runs
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(x)
        (y.float() * gy * scale).sum().backward()
        g = {"x": x.grad.float() / scale}
        g.update({k: p.grad.float() / scale for k, p in m.named_parameters() if p.grad is not None})
        grads.append(g)
    for k, a in grads[0].items():
        b = grads[1][k]
        assert torch.isfinite(b).all(), k
        err = (a - b).abs().max().item()
        ref = a.abs().max().item()
        assert err <= 2e-2 * ref + 1e-12, (k, err, ref)
