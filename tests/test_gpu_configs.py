"""-m gpu: the other workload geometries of BASELINE.json / SURVEY.md 8(d) -- cfg2 (DeiT-tiny-p16,
N = 196), cfg4 (PvT stages, N up to 9216, d = 64) and cfg5 (1-D N = 4096, h = 8, pad mask) --
at sizes the golden fixtures do not reach.  For each: the convexity identity on the whole batch
(every output is a normalised combination of the values) and forward + input gradient of one batch
element against the CPU oracle.  Module level, i.e. through the C ABI."""
import os
import sys
import warnings

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-attention_amd"), os.path.join(ROOT, "tests")]

EVA2D = dict(attn_2d=True, use_rpe=True, adaptive_proj="default")
CONFIGS = {
    # cfg2: DeiT-tiny-p16
    "cfg2_eva": ("eva", (128, 14, 14, 192), dict(dim=192, num_heads=3, window_size=7, num_landmarks=49, **EVA2D), None),
    "cfg2_lara": ("lara", (128, 14, 14, 192), dict(dim=192, num_heads=3, num_landmarks=49, proposal_gen="pool-mixed",
                                                    mis_type="mis-opt", alpha_coeff=2.0), None),
    "cfg2_softmax": ("softmax", (128, 14, 14, 192), dict(dim=192, num_heads=3), None),
    # cfg4: PvT-b2 stages at 384^2 (h, N) = (1, 9216), (2, 2304), (5, 576), window 8, 36 landmarks; softmax (8, 144)
    "cfg4_eva_s1": ("eva", (8, 96, 96, 64), dict(dim=64, num_heads=1, window_size=8, num_landmarks=36, **EVA2D), None),
    "cfg4_eva_s2": ("eva", (8, 48, 48, 128), dict(dim=128, num_heads=2, window_size=8, num_landmarks=36, **EVA2D), None),
    "cfg4_eva_s3": ("eva", (8, 24, 24, 320), dict(dim=320, num_heads=5, window_size=8, num_landmarks=36, **EVA2D), None),
    "cfg4_softmax_s4": ("softmax", (8, 12, 12, 512), dict(dim=512, num_heads=8), None),
    # cfg5: 1-D N = 4096, h = 8, 10 % trailing pads in the second row
    "cfg5_lara_L16": ("lara", (4, 4096, 512), dict(dim=512, num_heads=8, num_landmarks=16, proposal_gen="adaptive-1d",
                                                   mis_type="mis-opt"), [0, 410, 0, 17]),
    "cfg5_lara_L49": ("lara", (2, 4096, 512), dict(dim=512, num_heads=8, num_landmarks=49, proposal_gen="adaptive-1d",
                                                   mis_type="mis-opt"), None),
    "cfg5_eva_1d": ("eva", (4, 4096, 512), dict(dim=512, num_heads=8, window_size=16, attn_2d=False, use_t5_rpe=True,
                                                overlap_window=True, num_landmarks=8, adaptive_proj="default"), [0, 410, 0, 17]),
    "cfg5_local_1d": ("local", (4, 4096, 512), dict(dim=512, num_heads=8, window_size=16, attn_2d=False, use_rpe=True),
                      [0, 410, 0, 17]),
    "cfg5_performer": ("performer", (4, 4096, 512), dict(dim=512, num_heads=8, approx_attn_dim=64, proj_method="favorp"),
                       [0, 410, 0, 17]),
    "cfg2_scatterbrain": ("scatterbrain", (16, 14, 14, 192), dict(dim=192, num_heads=3, window_size=7, attn_2d=True, use_rpe=True,
                                                                  approx_attn_dim=64), None),
    "cfg5_scatterbrain": ("scatterbrain", (2, 1000, 512), dict(dim=512, num_heads=8, window_size=16, attn_2d=False, use_rpe=True,
                                                               approx_attn_dim=64), [0, 130]),
    # windows larger than one LDS image in backward: query blocks, run one after the other because the
    # windows overlap or the causal key lists do not apply (ea_window_bwd_query_blocks > 1, acc_slices = 1)
    "big_eva_1d_overlap": ("eva", (2, 512, 512), dict(dim=512, num_heads=8, window_size=128, attn_2d=False, use_t5_rpe=True,
                                                      overlap_window=True, num_landmarks=8, adaptive_proj="default"), [0, 50]),
    "big_eva_1d_w256": ("eva", (2, 512, 512), dict(dim=512, num_heads=8, window_size=256, attn_2d=False, use_rpe=True,
                                                   num_landmarks=8, adaptive_proj="no-ln"), [0, 50]),
    "big_local_1d": ("local", (2, 1024, 256), dict(dim=256, num_heads=4, window_size=128, attn_2d=False, use_rpe=True,
                                                   overlap_window=True), [0, 100]),
}


def _build(attn, args):
    import efficient_attention as ea
    torch.manual_seed(21)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = ea.AttentionFactory.build_attention(attn, dict(args)).cuda()
    m.eval()
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.02 * torch.randn_like(p))
    return m


def _xscale(attn):
    # Performer divides by clamp(phi(q).sum_k phi(k), 1e-2) (kernelized_attention.py:55): with unit-
    # variance inputs and untrained weights at N = 4096 the clamp is active for some queries, where
    # the map is neither convex in v nor differentiable -- test it away from the kink
    return 0.25 if attn == "performer" else 1.0


def _mask(shape, pads):
    if pads is None:
        return None
    n = 1
    for s in shape[1:-1]:
        n *= s
    mask = torch.zeros(shape[0], n, dtype=torch.bool, device="cuda")
    for b, k in enumerate(pads):
        if k:
            mask[b, n - k:] = True
    return mask


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CONFIGS))
def test_constant_values_pass_through(name):
    attn, shape, args, pads = CONFIGS[name]
    m = _build(attn, args)
    C = shape[-1]
    cvec = torch.linspace(-1.0, 1.0, C, device="cuda")
    with torch.no_grad():
        m.qkv.weight[2 * C:].zero_()
        m.qkv.bias[2 * C:].copy_(cvec)
        m.proj.weight.copy_(torch.eye(C, device="cuda"))
        m.proj.bias.zero_()
        x = _xscale(attn) * torch.randn(*shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
        mask = _mask(shape, pads)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = (m(x, mask) if mask is not None else m(x)).float()
    assert torch.isfinite(y).all()
    ref = cvec.to(torch.bfloat16).float().expand_as(y)
    assert (y - ref).abs().max().item() <= 2e-2


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CONFIGS))
def test_one_element_matches_oracle(name):
    import oracle
    from gpu_checks import MODULE_TOL, LARA_TOL, SCATTER_TOL
    from util import scaled_err
    attn, shape, args, pads = CONFIGS[name]
    m = _build(attn, args)
    gen = torch.Generator(device="cuda").manual_seed(5)
    x = (_xscale(attn) * torch.randn(*shape, device="cuda", generator=gen)).requires_grad_(True)
    gy = torch.randn(*shape, device="cuda", generator=gen)
    mask = _mask(shape, pads)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = m(x, mask) if mask is not None else m(x)
    (y.float() * gy).sum().backward()
    b = 1 if pads is not None else shape[0] // 2             # the element with the long pad run, if any
    sl = slice(b, b + 1)
    params = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    xr = x.detach()[sl].cpu().requires_grad_(True)
    mr = None if mask is None else mask[sl].cpu()
    ref = oracle.module_forward(attn, dict(args), params, xr, mr, training=False)
    (ref * gy[sl].cpu()).sum().backward()
    from gpu_checks import tol_for
    tol = tol_for(attn, "bf16", "test_gpu_configs")
    for what, got, want in (("y", y.detach().float()[sl].cpu(), ref.detach()), ("dx", x.grad[sl].cpu(), xr.grad)):
        e = scaled_err(got.numpy(), want.numpy())
        assert e[0] <= tol[0] and e[1] <= tol[1], (name, what, e)
