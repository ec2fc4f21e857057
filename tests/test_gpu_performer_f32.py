"""-m gpu: the Performer baseline in exact fp32 arithmetic (ea_performer_f32_*, VERDICT r03 missing #1 / next #7).
The reference forces full precision in its linear attention even under AMP (kernelized_attention.py:116-121,343-345) and
computes fp32 outside autocast (abstract_attention.py:120-133); these tests hold the HIP path to fp32 tolerances."""
import glob
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "efficient-attention_amd"), HERE, os.path.join(HERE, "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

# the module-level tests need the default (exact fp32) core; EA_PERFORMER_16BIT=1 selects the 16-bit kernels for the process
NEEDS_F32_DEFAULT = pytest.mark.skipif(os.environ.get("EA_PERFORMER_16BIT", "0") == "1",
                                       reason="EA_PERFORMER_16BIT=1: the 16-bit Performer kernels are the process default")
# (performer_2d_d32: head_dim 32 runs on the 16-bit kernels -- the exact-fp32 core is built for head_dim 64 -- and is held to the
#  bf16 / fp16 bounds by tests/test_gpu_modules.py; in fp32 outside autocast it takes the documented rounding path)
PERFORMER_FIXTURES = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(HERE, "golden", "performer_*.npz"))
                            if not f.endswith("_d32.npz"))


def _oracle(q, k, v, mask, W):
    import torch
    from oracle import attention as oa
    return oa.performer_core(q, k, v, mask, W)


@pytest.mark.gpu
@pytest.mark.parametrize("B,h,N,m,io,masked", [(2, 3, 784, 64, "fp32", False), (2, 2, 200, 64, "fp32", True), (1, 8, 4096, 64, "fp32", True),
                                               (3, 3, 196, 32, "fp32", False), (2, 2, 333, 96, "fp32", True),
                                               (2, 3, 784, 64, "bf16", False), (2, 2, 200, 64, "fp16", True)])
def test_performer_f32_core_matches_fp64_oracle(B, h, N, m, io, masked):
    """out, dq, dk, dv of the fp32-arithmetic core against the oracle evaluated in fp64 on the SAME inputs (fp32 inputs, or
    16-bit inputs whose values are exact in fp32): max |err| <= 1e-4 of the tensor's largest value (observed ~1e-6; 16-bit
    I/O: one rounding of the outputs)."""
    import torch
    from efficient_attention import _ops
    td = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[io]
    g = torch.Generator(device="cuda").manual_seed(N + m)
    qkv = (torch.randn(B, N, 3, h, 64, device="cuda", generator=g) * 0.7).to(td).requires_grad_(True)
    W = torch.randn(h, m, 64, device="cuda", generator=g)
    mask = None
    if masked:
        mask = torch.zeros(B, N, dtype=torch.bool, device="cuda")
        mask[0, N - N // 5:] = True
        mask[-1, 3:40] = True
    dout = torch.randn(B, N, h, 64, device="cuda", generator=g).to(td)
    out = _ops.PerformerF32Fn.apply(qkv, _ops._mask_u8(mask, B, N, qkv.device), W)
    assert out.dtype == td
    out.backward(dout)
    q64 = qkv.detach().double().cpu().requires_grad_(True)
    q_, k_, v_ = (q64[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    ref = _oracle(q_, k_, v_, None if mask is None else mask.cpu(), W.double().cpu())            # [B,h,N,d]
    ref.backward(dout.double().cpu().permute(0, 2, 1, 3))
    tol = 1e-4 if io == "fp32" else (2.0 ** -8 if io == "bf16" else 2.0 ** -10)
    o_err = (out.detach().double().cpu().permute(0, 2, 1, 3) - ref.detach()).abs().max() / ref.detach().abs().max()
    assert float(o_err) <= tol, float(o_err)
    gref = q64.grad
    for i, nm in enumerate("qkv"):
        a, b = qkv.grad[:, :, i].double().cpu(), gref[:, :, i]
        assert float((a - b).abs().max() / b.abs().max()) <= tol, (nm, float((a - b).abs().max() / b.abs().max()))


@NEEDS_F32_DEFAULT
@pytest.mark.gpu
@pytest.mark.parametrize("name", PERFORMER_FIXTURES)
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_performer_module_fp32_outside_autocast_matches_reference(name, mode):
    """The product module called in fp32 WITHOUT autocast -- fp32 projections (library GEMM), the exact-fp32 HIP core --
    against the fp32 reference's golden vectors at fp32 tolerances, including the discontinuous-clamp case that needs a
    14 % band with 16-bit operands (tests/gpu_checks.py CASE_TOL)."""
    import torch
    from gpu_checks import check_module_case
    errs = check_module_case(name, mode, dtype=torch.float32, tol=(2e-4, 1e-4))
    assert errs


@NEEDS_F32_DEFAULT
@pytest.mark.gpu
def test_performer_autocast_uses_the_fp32_core_by_default():
    import warnings
    import torch
    import efficient_attention as ea
    from efficient_attention import _ops
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = ea.AttentionFactory.build_attention("performer", dict(dim=192, num_heads=3, approx_attn_dim=64)).cuda()
    x = torch.randn(2, 14, 14, 192, device="cuda", requires_grad=True)
    calls = []
    real = _ops.performer_f32_fwd
    _ops.performer_f32_fwd = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(x)
    finally:
        _ops.performer_f32_fwd = real
    names = set()
    fn = y.grad_fn
    stack = [fn]
    while stack:
        f = stack.pop()
        if f is None:
            continue
        names.add(type(f).__name__)
        stack.extend(n for n, _ in f.next_functions)
    # the fp32 core ran: as its own node (PerformerF32Fn) or inside the single node of the module (CoreModuleFn, PerformerCore)
    assert len(calls) == 1 and any(n.startswith(("PerformerF32Fn", "CoreModuleFn")) for n in names), (calls, names)
    assert not _ops.PERFORMER_16BIT


@NEEDS_F32_DEFAULT
@pytest.mark.gpu
@pytest.mark.parametrize("shape,args", [((32, 28, 28, 192), dict(dim=192, num_heads=3, approx_attn_dim=64)),
                                        ((2, 4096, 512), dict(dim=512, num_heads=8, approx_attn_dim=64))])
def test_performer_fp32_fullsize_at_unit_inputs_matches_oracle(shape, args):
    """the full-size Performer runs were only pinned at inputs scaled by 0.25 (away from the clamp of the
    normaliser, whose derivative is discontinuous).  In fp32 outside autocast -- fp32 projections, the exact-fp32 HIP core --
    the whole batch at UNIT inputs, training mode with shared feature draws, agrees with the oracle on y, dx and every
    parameter gradient at fp32 tolerances."""
    import contextlib
    import warnings
    import torch
    import efficient_attention as ea
    import oracle
    from util import scaled_err
    torch.manual_seed(31)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = ea.AttentionFactory.build_attention("performer", dict(args)).cuda()
    m.train()
    gen = torch.Generator(device="cuda").manual_seed(6)
    x = torch.randn(*shape, device="cuda", generator=gen).requires_grad_(True)
    gy = torch.randn(*shape, device="cuda", generator=gen)
    real = torch.randn
    draws = []

    def randn(*size, **kw):
        if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
            size = tuple(size[0])
        t = real(tuple(size), generator=torch.Generator().manual_seed(77 + len(draws)))
        draws.append(t)
        return t.to(device=kw.get("device", "cuda"), dtype=kw.get("dtype") or torch.float32)
    torch.randn = randn
    try:
        y = m(x)
    finally:
        torch.randn = real
    assert y.dtype == torch.float32
    (y * gy).sum().backward()
    params = {k: v.detach().float().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in m.state_dict().items()}
    xr = x.detach().cpu().requires_grad_(True)
    it = iter(draws)
    ref = oracle.module_forward("performer", dict(args), params, xr, None, training=True, noise_fn=lambda shp: next(it))
    (ref * gy.cpu()).sum().backward()
    pairs = [("y", y.detach().cpu().numpy(), ref.detach().numpy()), ("dx", x.grad.cpu().numpy(), xr.grad.numpy())]
    for k, p_ in m.named_parameters():
        if params[k].grad is not None:
            pairs.append(("d" + k, p_.grad.cpu().numpy(), params[k].grad.numpy()))
    bad = {}
    for what, got, want in pairs:
        e = scaled_err(got, want)
        if not (e[0] <= 5e-4 and e[1] <= 2e-4):
            bad[what] = e
    assert not bad, bad


@pytest.mark.gpu
def test_scatterbrain_fp32_input_outside_autocast_head_dim_64_runs():
    """ScatterBrain inherits KernelizedAttention.project_qkv through the MRO; the fp32 pass-through of that method
    is for the Performer core only -- with head_dim 64 and the default 64 features ScatterBrain on plain fp32 input used to hand
    fp32 rows to the 16-bit window kernels ('attention cores take bf16 or fp16 tensors')."""
    import warnings
    import torch
    import efficient_attention as ea
    from oracle import attention as oa
    torch.manual_seed(3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = ea.AttentionFactory.build_attention("scatterbrain", dict(dim=128, num_heads=2, window_size=4, approx_attn_dim=64)).cuda().eval()
        x = torch.randn(2, 32, 128, device="cuda", requires_grad=True)
        y = m(x)
        y.float().sum().backward()
    assert y.shape == x.shape and torch.isfinite(y).all() and torch.isfinite(x.grad).all()
    args = oa.default_args("scatterbrain")
    args.update(dim=128, num_heads=2, window_size=4, approx_attn_dim=64)
    params = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    ref = oa.module_forward("scatterbrain", args, params, x.detach().float().cpu(), training=False)
    err = (y.detach().float().cpu() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 8e-2, err
