"""-m gpu: the fp32-FAITHFUL cores (round 5; csrc/ea_f32_attn.hip, efficient_attention/_f32.py; VERDICT r04 missing #1).
Outside torch.autocast the reference computes attention in fp32 (abstract_attention.py:120-133, local_attention.py:134-182,
eva.py:138-233).  The softmax baseline, LocalAttention and EVA called the same way keep fp32 end to end and are held to fp32
tolerances here: every softmax_* / local_* / eva_* golden vector of the reference at 2e-4 (max) / 1e-4 (rms), forward, input
gradient and every parameter gradient; the gathered-attention kernels themselves against the oracle in fp64."""
import glob
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "efficient-attention_amd"), HERE, os.path.join(HERE, "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

F32_FIXTURES = sorted(os.path.basename(f)[:-4] for pat in ("softmax_*", "local_*", "eva_*", "lara_*", "ra_*", "causal_eva_*")
                      for f in glob.glob(os.path.join(HERE, "golden", pat + ".npz")))
F32_TOL = (2e-4, 1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("name", F32_FIXTURES)
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_module_fp32_outside_autocast_matches_reference(name, mode):
    """The product module in fp32 WITHOUT autocast (fp32 library GEMMs for the two Linear layers, ea_f32_attn_* for the core)
    against the fp32 reference's own outputs: y, dx and every parameter gradient at fp32 tolerances -- and the fp32 path is what
    ran (no rounding warning, the fp32 entry points were called)."""
    import warnings
    import torch
    from efficient_attention import _native as nv
    from gpu_checks import check_module_case
    calls = []
    real = nv.call
    nv.call = lambda nm, *a: (calls.append(nm), real(nm, *a))[1]
    try:
        from efficient_attention import _ops
        _ops._FP32_WARNED[0] = False
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            errs = check_module_case(name, mode, dtype=torch.float32, tol=F32_TOL)
        assert not [w for w in rec if "rounded to bf16" in str(w.message)]     # _ops.to_io_dtype did not round anything
    finally:
        nv.call = real
    assert errs and "ea_f32_attn_fwd" in calls and "ea_f32_attn_bwd" in calls, sorted(set(calls))
    assert not [c for c in calls if c.startswith(("ea_window", "ea_softmax", "ea_eva_", "ea_lara_", "ea_rows_mlp"))], sorted(set(calls))


def _rel(a, b):
    return float((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.gpu
@pytest.mark.parametrize("B,h,N,d,masked,drop", [(2, 3, 200, 64, True, False), (1, 2, 4096, 64, True, False), (2, 2, 96, 32, False, True),
                                                  (1, 1, 130, 128, True, False)])
def test_f32_softmax_core_matches_fp64_oracle(B, h, N, d, masked, drop):
    import torch
    from efficient_attention import _f32
    from oracle import attention as oa
    torch.manual_seed(N + d)
    qkv = torch.randn(B, N, 3, h, d, device="cuda", requires_grad=True)
    mask = None
    if masked:
        mask = torch.zeros(B, N, dtype=torch.bool)
        mask[0, N - N // 5:] = True
    keep = None
    if drop:
        ld = -(-N // 64) * 64
        keep = (torch.rand(B, h, N, ld, device="cuda") > 0.2).to(torch.uint8)
    out = _f32.softmax_core(qkv, None if mask is None else mask.cuda().to(torch.uint8), keep, 1.25 if drop else 1.0)
    gy = torch.randn_like(out)
    (out * gy).sum().backward()
    q64 = qkv.detach().double().cpu().requires_grad_(True)
    q, k, v = [q64[:, :, i].permute(0, 2, 1, 3) for i in range(3)]
    ref = oa.softmax_core(q, k, v, mask, None, None if keep is None else keep[..., :N].double().cpu(), 0.2 if drop else 0.0)
    (ref * gy.double().cpu().permute(0, 2, 1, 3)).sum().backward()
    assert _rel(out.detach().permute(0, 2, 1, 3), ref.detach()) < 2e-5
    assert _rel(qkv.grad, q64.grad) < 5e-5


@pytest.mark.gpu
@pytest.mark.parametrize("attn_2d,shape,w,e,d,masked", [(True, (14, 14), 7, 0, 64, False), (True, (14, 14), 7, 3, 64, False),
                                                         (False, (50,), 4, 2, 32, True), (False, (256,), 128, 0, 128, True),
                                                         (True, (16, 16), 8, 4, 64, False)])
def test_f32_local_and_eva_cores_match_fp64_oracle(attn_2d, shape, w, e, d, masked):
    """local_core and eva_core (masked chunk means, beta with the key-norm term and zeroed masked values, windows with the
    control-variate columns, bias) against the oracle in fp64: outputs and the gradients of q, k, v, the bias and the mu rows."""
    import math
    import torch
    from efficient_attention import _f32
    from oracle import attention as oa
    torch.manual_seed(w * 7 + e)
    B, h = 2, 2
    N = int(math.prod(shape))
    if not attn_2d and N % w:
        N = -(-N // w) * w                                  # EVA pads x first (eva.py:127-136): cores see multiples of w
        shape = (N,)
    Wq, Wk = (w * w, (w + 2 * e) ** 2) if attn_2d else (w, w + 2 * e)
    qkv = torch.randn(B, N, 3, h, d, device="cuda", requires_grad=True)
    bias = (0.3 * torch.randn(h, Wq, Wk, device="cuda")).requires_grad_(True)
    mask = None
    if masked:
        mask = torch.zeros(B, N, dtype=torch.bool)
        mask[1, N - N // 3:] = True
    m8 = None if mask is None else mask.cuda().to(torch.uint8)
    gy = torch.randn(B, N, h, d, device="cuda")
    q64 = qkv.detach().double().cpu().requires_grad_(True)
    b64 = bias.detach().double().cpu().requires_grad_(True)
    q, k, v = [q64[:, :, i].permute(0, 2, 1, 3) for i in range(3)]
    # local
    out = _f32.local_core(qkv, bias, m8, attn_2d, shape, w, e)
    (out * gy).sum().backward()
    ref = oa.local_core(q, k, v, mask, attn_2d, w, e, b64)
    (ref * gy.double().cpu().permute(0, 2, 1, 3)).sum().backward()
    assert _rel(out.detach().permute(0, 2, 1, 3), ref.detach()) < 2e-5
    assert _rel(qkv.grad, q64.grad) < 5e-5 and _rel(bias.grad, b64.grad) < 5e-5
    # EVA: L landmarks chosen so that the chunk side divides the grid
    qkv.grad = None; bias.grad = None; q64.grad = None; b64.grad = None
    L = (4 if shape[0] % 4 == 0 else 1) if not attn_2d else 4
    r = int(math.sqrt(N // L)) if attn_2d else N // L
    A = torch.randn(d, d, device="cuda", requires_grad=True)
    noise = torch.randn(B, h, (shape[0] // r) * (shape[1] // r) if attn_2d else -(-N // r), d, device="cuda")

    def mu_fn(qm, km, A=A):
        rk = km @ A
        return rk, 0.5 * (qm + rk)
    out = _f32.eva_core(qkv, bias, noise, m8, attn_2d, shape, w, e, r, mu_fn)
    (out * gy).sum().backward()
    A64 = A.detach().double().cpu().requires_grad_(True)
    q, k, v = [q64[:, :, i].permute(0, 2, 1, 3) for i in range(3)]
    ref = oa.eva_core(q, k, v, mask, attn_2d, shape, w, e, L, lambda qm, km: mu_fn(qm, km, A64), noise.double().cpu(), b64)
    (ref * gy.double().cpu().permute(0, 2, 1, 3)).sum().backward()
    assert _rel(out.detach().permute(0, 2, 1, 3), ref.detach()) < 2e-5
    assert _rel(qkv.grad, q64.grad) < 1e-4 and _rel(bias.grad, b64.grad) < 1e-4 and _rel(A.grad, A64.grad) < 1e-4


@pytest.mark.gpu
def test_other_variants_still_round_fp32_input_with_a_warning():
    """ScatterBrain has no fp32-operand core yet: fp32 input outside autocast is rounded to bf16 and the caller is
    told (once per process: _ops._FP32_WARNED is reset here)."""
    import warnings
    import torch
    import efficient_attention as ea
    from efficient_attention import _ops
    m = ea.AttentionFactory.build_attention("scatterbrain", dict(dim=128, num_heads=2, window_size=4)).cuda().eval()
    _ops._FP32_WARNED[0] = False
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        y = m(torch.randn(2, 32, 128, device="cuda"))
    assert y.dtype == torch.float32 and any("rounded to bf16" in str(w.message) for w in rec)
