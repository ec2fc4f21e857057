"""Hardware assumptions of every kernel (MFMA operand layout, ds_read_b64_tr_b16 mapping)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_probe_primitives():
    exe = os.path.join(ROOT, "efficient-attention_amd", "lib", "probe_primitives")
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0 and "PROBE ALL OK" in r.stdout, r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols", [(100352, 576), (100352, 192), (1000, 64), (37, 8), (5000, 2048), (4096, 2560), (300, 4104)])
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_bias_grad_matches_fp64_column_sum(rows, cols, dtype):
    """ea_bias_grad (projection bias gradient) against an fp64 column sum of the same bf16/fp16 data."""
    import torch
    from efficient_attention import _ops
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    g = torch.Generator(device="cuda").manual_seed(rows + cols)
    dy = (torch.randn(rows, cols, device="cuda", generator=g) + 0.25).to(td)
    got = _ops.bias_grad(dy)
    ref = dy.double().sum(0)
    assert got.dtype == torch.float32 and got.shape == (cols,)
    # fp32 accumulation of <= 1e5 terms of magnitude ~1: error well below 1e-5 relative to sum |x|
    tol = 2e-6 * dy.double().abs().sum(0)
    assert bool(((got.double() - ref).abs() <= tol + 1e-6).all())
    again = _ops.bias_grad(dy)
    assert torch.equal(got, again)          # fixed-order two-stage sum: deterministic


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols", [(384, 8192), (384, 384), (7, 5), (1000, 33)])
def test_colsum_f32_matches_fp64(rows, cols):
    import torch
    from efficient_attention import _ops
    g = torch.Generator(device="cuda").manual_seed(rows * 31 + cols)
    x = torch.randn(rows, cols, device="cuda", generator=g)
    got = _ops.colsum_f32(x)
    ref = x.double().sum(0)
    assert torch.allclose(got.double(), ref, rtol=0, atol=2e-6 * float(x.abs().sum(0).max()) + 1e-6)
    assert torch.equal(got, _ops.colsum_f32(x))


@pytest.mark.gpu
@pytest.mark.parametrize("with_a", [True, False])
def test_slice_sum(with_a):
    import ctypes
    import torch
    from efficient_attention import _native as nv
    BH, S, n = 12, 3, 49 * 64
    g = torch.Generator(device="cuda").manual_seed(5)
    a = torch.randn(BH, n, device="cuda", generator=g)
    p = torch.randn(BH, S, n, device="cuda", generator=g)
    out = torch.empty(BH, n, device="cuda")
    nv.call("ea_slice_sum", BH, S, n, 0.125, nv.ptr(a) if with_a else None, nv.ptr(p), nv.ptr(out), nv.stream())
    ref = 0.125 * ((a if with_a else 0) + p.sum(1))
    assert torch.allclose(out, ref, rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("ln", [True, False])
@pytest.mark.parametrize("sides", [1, 2])
@pytest.mark.parametrize("R,D", [(9216, 128), (1000, 64), (37, 32), (18816, 64)])
def test_rows_mlp_matches_fp64(R, D, sides, ln):
    """ea_rows_mlp_fwd/bwd (the mu networks: per-row Linear [+ LayerNorm], exact fp32 MFMA) against
    torch in fp64: outputs, input gradients and every parameter gradient."""
    import ctypes
    import torch
    import torch.nn.functional as F
    from efficient_attention import _native as nv
    from efficient_attention import _ops
    gen = torch.Generator(device="cuda").manual_seed(R + D + sides)

    def rnd(*shape, scale=1.0, shift=0.0):
        return (torch.randn(*shape, device="cuda", generator=gen) * scale + shift).contiguous()
    xs = [rnd(R, D) for _ in range(sides)]
    Ws = [rnd(D, D, scale=D ** -0.5) for _ in range(sides)]
    bs = [rnd(D, scale=0.3) for _ in range(sides)]
    gs = [rnd(D, scale=0.2, shift=1.0) for _ in range(sides)]
    cs = [rnd(D, scale=0.3) for _ in range(sides)]
    dys = [rnd(R, D) for _ in range(sides)]

    def two(ts):
        ts = list(ts) + [None]
        return [nv.ptr(ts[0]), nv.ptr(ts[1])]
    ys = [torch.empty(R, D, device="cuda") for _ in range(sides)]
    zhat = torch.empty(sides, R, D, device="cuda") if ln else None
    rstd = torch.empty(sides, R, device="cuda") if ln else None
    nv.call("ea_rows_mlp_fwd", R, D, sides, int(ln), *two(xs), *two(Ws), *two(bs), *two(gs if ln else [None]),
            *two(cs if ln else [None]), *two(ys), nv.ptr(zhat), nv.ptr(rstd), nv.stream())
    parts = nv.lib().ea_rows_mlp_parts(R, D)
    planes = 3 if ln else 1
    dxs = [torch.empty(R, D, device="cuda") for _ in range(sides)]
    feed = torch.empty(R, planes, sides, D, device="cuda")
    dWp = torch.empty(parts, sides, D, D, device="cuda")
    nv.call("ea_rows_mlp_bwd", R, D, sides, int(ln), *two(dys), *two(xs), *two(Ws), *two(gs if ln else [None]),
            nv.ptr(zhat), nv.ptr(rstd), *two(dxs), nv.ptr(feed), nv.ptr(dWp), nv.stream())
    dW = _ops.colsum_f32(dWp.view(parts, -1)).view(sides, D, D)
    vec = _ops.colsum_f32(feed.view(R, -1)).view(planes, sides, D)

    def close(got, ref, what):
        err = (got.double() - ref).abs().max().item()
        assert err <= 2e-5 * max(ref.abs().max().item(), 1.0) * (1 + R ** 0.5 * 0.02), (what, err, ref.abs().max().item())
    for s in range(sides):
        x = xs[s].double().requires_grad_(True)
        W, b = Ws[s].double().requires_grad_(True), bs[s].double().requires_grad_(True)
        g, c = gs[s].double().requires_grad_(True), cs[s].double().requires_grad_(True)
        y = F.linear(x, W, b)
        if ln:
            y = F.layer_norm(y, (D,), g, c, 1e-5)
        y.backward(dys[s].double())
        close(ys[s], y.detach(), "y")
        close(dxs[s], x.grad, "dx")
        close(dW[s], W.grad, "dW")
        close(vec[0, s], b.grad, "db")
        if ln:
            close(vec[1, s], g.grad, "dgamma")
            close(vec[2, s], c.grad, "dbeta")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("rows,M,K", [(100352, 576, 192), (100352, 192, 192), (25088, 576, 192), (4096, 1536, 512),
                                      (1000, 64, 64), (777, 128, 320), (294912, 192, 64), (65, 960, 320),
                                      (8192, 2560, 512), (3000, 512, 512), (700, 768, 256), (1025, 256, 192),
                                      (18432, 960, 320), (18432, 320, 320), (5001, 640, 640), (333, 128, 640)])
def test_wgrad_matches_fp64(rows, M, K, dtype):
    """ea_wgrad (weight + bias gradient of a projection in one pass) against fp64 on the same bf16/fp16 data:
    fp32 accumulation of exact products -> error far below one operand ulp of the result; deterministic."""
    import torch
    from efficient_attention import _ops
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    g = torch.Generator(device="cuda").manual_seed(rows + M + K)
    dy = (torch.randn(rows, M, device="cuda", generator=g) * 0.5 + 0.1).to(td)
    x = torch.randn(rows, K, device="cuda", generator=g).to(td)
    assert _ops.wgrad_supported(dy, x)
    dw, db = _ops.wgrad(dy, x, True)
    assert dw.shape == (M, K) and db.shape == (M,) and dw.dtype == torch.float32
    ref = dy.double().t() @ x.double()
    refb = dy.double().sum(0)
    scale = (dy.double().abs().t() @ x.double().abs())
    assert bool(((dw.double() - ref).abs() <= 4e-6 * scale + 1e-6).all()), float((dw.double() - ref).abs().max())
    assert bool(((db.double() - refb).abs() <= 4e-6 * dy.double().abs().sum(0) + 1e-6).all())
    dw2, db2 = _ops.wgrad(dy, x, True)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("rows,K,NO,a_f32,y_f32", [
    (100352, 192, 576, 1, 0), (100352, 192, 192, 0, 0), (25088, 192, 768, 1, 0), (1001, 64, 128, 1, 1),
    (50017, 128, 384, 0, 0), (30001, 256, 256, 1, 1), (77, 192, 576, 1, 0), (1, 64, 64, 0, 1), (4097, 256, 512, 0, 0)])
def test_linear_matches_fp64(rows, K, NO, a_f32, y_f32, dtype):
    """ea_linear (the qkv / output projection as a streaming kernel) against fp64 on the same rounded operands: fp32
    accumulation of exact products, one rounding of the result; an fp32 input is rounded exactly like .to(dtype) and
    the rounded copy handed back is that tensor."""
    import torch
    from efficient_attention import _ops
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    g = torch.Generator(device="cuda").manual_seed(rows + K + NO)
    a = torch.randn(rows, K, device="cuda", generator=g)
    a = a if a_f32 else a.to(td)
    w = (torch.randn(NO, K, device="cuda", generator=g) * K ** -0.5).to(td)
    b = torch.randn(NO, device="cuda", generator=g)
    assert _ops.ea_linear_supported(a, w)
    y, ac = _ops.ea_linear(a, w, b, torch.float32 if y_f32 else td, want_cast=bool(a_f32))
    a16 = a.to(td)
    if a_f32:
        assert torch.equal(ac, a16)
    ref = a16.double() @ w.double().t() + b.to(td).double()
    mag = a16.double().abs() @ w.double().abs().t() + b.double().abs()
    tol = 4e-6 * mag + (0 if y_f32 else 1) * (2.0 ** (-8 if dtype == "bf16" else -11)) * ref.abs() + 1e-6
    assert y.dtype == (torch.float32 if y_f32 else td)
    assert bool(((y.double() - ref).abs() <= tol).all()), float((y.double() - ref).abs().max())
    y2, _ = _ops.ea_linear(a, w, b, torch.float32 if y_f32 else td, want_cast=False)
    assert torch.equal(y, y2)
    # strided rows (a column slice of a wider tensor)
    wide = torch.randn(rows, K + 64, device="cuda", generator=g).to(a.dtype)
    av = wide[:, :K]
    assert _ops.ea_linear_supported(av, w)
    y3, _ = _ops.ea_linear(av, w, None, td)
    ref3 = av.to(td).double() @ w.double().t()
    mag3 = av.to(td).double().abs() @ w.double().abs().t()
    assert bool(((y3.double() - ref3).abs() <= 4e-6 * mag3 + 2.0 ** (-8 if dtype == "bf16" else -11) * ref3.abs() + 1e-6).all())


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("rows,w_f32,dx_f32,wide", [(100352, 0, 1, 0), (100352, 1, 1, 0), (25088, 0, 0, 0), (1000, 1, 0, 1),
                                                      (33, 0, 1, 1), (1, 1, 1, 0)])
def test_linear_dgrad_matches_fp64(rows, w_f32, dx_f32, wide, dtype):
    """ea_linear_dgrad (round 5: the qkv projection's input gradient dx = dqkv W, 576 -> 192, weight resident in registers;
    abstract_attention.py:72-78 differentiated) against fp64 on the same rounded operands: fp32 accumulation of exact
    products (one rounding when the result is 16-bit), from the fp32 master weight (rounded like .to(dtype)) or its 16-bit copy,
    on contiguous and strided gradient rows, bit-reproducible; _ops.qkv_dgrad falls back to the library outside 576 x 192."""
    import torch
    from efficient_attention import _ops, _native as nv
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    g = torch.Generator(device="cuda").manual_seed(rows + 7 * w_f32 + dx_f32)
    buf = (0.5 * torch.randn(rows, 576 + (64 if wide else 0), device="cuda", generator=g)).to(td)
    dy = buf[:, :576]
    w = torch.randn(576, 192, device="cuda", generator=g) * 0.05
    w16 = w.to(td)
    xd = torch.float32 if dx_f32 else td
    assert nv.lib().ea_linear_dgrad_supported(192, 576) and not nv.lib().ea_linear_dgrad_supported(512, 1536)
    old = _ops.DGRAD_RS_MIN_ROWS
    _ops.DGRAD_RS_MIN_ROWS = 1
    try:
        calls = []
        real = nv.call_as
        nv.call_as = lambda label, name, *a: (calls.append(name), real(label, name, *a))[1]
        try:
            dx = _ops.qkv_dgrad(dy, w, None if w_f32 else w16, xd)
            dx2 = _ops.qkv_dgrad(dy, w, None if w_f32 else w16, xd)
        finally:
            nv.call_as = real
        assert calls == ["ea_linear_dgrad"] * 2, calls
    finally:
        _ops.DGRAD_RS_MIN_ROWS = old
    assert dx.dtype == xd and tuple(dx.shape) == (rows, 192) and torch.equal(dx, dx2)
    ref = dy.double() @ w16.double()
    mag = dy.double().abs() @ w16.double().abs()
    tol = 4e-6 * mag + (0 if dx_f32 else 1) * (2.0 ** (-8 if dtype == "bf16" else -11)) * ref.abs() + 1e-6
    assert bool(((dx.double() - ref).abs() <= tol).all()), float((dx.double() - ref).abs().max())
    # another width: the library GEMM takes it (no HIP entry for it), same contract
    w2 = torch.randn(576, 128, device="cuda", generator=g) * 0.05
    dx3 = _ops.qkv_dgrad(dy, w2, None, torch.float32)
    assert torch.allclose(dx3.double(), dy.double() @ w2.to(td).double(), rtol=1e-3, atol=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("B,H,W,r,C,has_t,dx_f32,w_f32", [(8, 28, 28, 4, 49, 1, 1, 0), (128, 14, 14, 2, 49, 1, 1, 1), (3, 28, 28, 4, 49, 0, 0, 0),
                                                          (5, 16, 16, 4, 16, 1, 0, 1), (300, 8, 8, 2, 16, 0, 1, 0),
                                                          # units that end in a half-filled tile (49 cells of two-cell tiles, 49
                                                          # cells of eight-cell tiles): the duplicate slots of round 6
                                                          (7, 28, 28, 4, 49, 0, 1, 1), (6, 14, 14, 2, 49, 1, 0, 0), (9, 14, 14, 2, 49, 0, 0, 1)])
def test_linear_dgrad_finish_equals_finish_then_dgrad(B, H, W, r, C, has_t, dx_f32, w_f32, dtype):
    """ea_linear_dgrad_finish (round 5: the last corrections of dq / dk -- lara.py:223 and the pooling backward of lara.py:43,48,
    145-151 / eva.py:178-181 -- fused into the input-gradient pass) against the two launches it replaces, ea_lara_bwd_finish
    followed by ea_linear_dgrad: the same arithmetic in the same order -> the corrected gradient rows and dx are BIT-identical."""
    import ctypes
    import torch
    from efficient_attention import _ops, _native as nv
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    g = torch.Generator(device="cuda").manual_seed(B * 31 + H + C)
    N, h, d = H * W, 3, 64
    L = (H // r) * (W // r)
    rn = lambda *shape, s=1.0: s * torch.randn(*shape, device="cuda", generator=g)   # noqa: E731
    qkv = rn(B, N, 3, h, d, s=0.5).to(td)
    dqkv = rn(B, N, 3, h, d, s=0.5).to(td)
    w = rn(576, 192, s=0.05)
    w16 = w.to(td)
    scale = d ** -0.5
    qbar = rn(B * h, C, d)
    uq = rn(B * h, C, d) if has_t else None
    q = qkv[:, :, 0].permute(0, 2, 1, 3).float()
    lse_t = torch.logsumexp(scale * torch.einsum("bhnd,bhcd->bhcn", q, qbar.view(B, h, C, d)), -1).reshape(B * h, C).contiguous()
    dpq, dpk = rn(B * h, L, d), rn(B * h, L, d)
    io = nv.io_dtype(qkv)
    geom = nv.ea_lara_geom(B, h, N, d, io, C, 0, 2.0, scale)
    # A: finish pass, then the plain input-gradient kernel
    da = dqkv.clone()
    qv, _, _ = _ops._qkv_views(qkv)
    dqv, dkv_, _ = _ops._qkv_views(da)
    tq, tdq, tdk = nv.t4(qv), nv.t4(dqv), nv.t4(dkv_)
    nv.call("ea_lara_bwd_finish", ctypes.byref(geom), ctypes.byref(tq), nv.ptr(qbar), nv.ptr(uq), nv.ptr(lse_t) if has_t else None,
            nv.ptr(dpq), nv.ptr(dpk), r, H, W, ctypes.byref(tdq), ctypes.byref(tdk), nv.stream())
    old = _ops.DGRAD_RS_MIN_ROWS
    _ops.DGRAD_RS_MIN_ROWS = 1
    try:
        xd = torch.float32 if dx_f32 else td
        dxa = _ops.qkv_dgrad(da.view(-1, 576), w, None if w_f32 else w16, xd)
        # B: one pass
        db = dqkv.clone()
        fin = dict(B=B, gh=H, gw=W, r=r, C=C, scale=scale, qbar=qbar if has_t else None, uq=uq, lse_t=lse_t if has_t else None,
                   dpq=dpq, dpk=dpk)
        dxb = _ops.qkv_dgrad_finish(db.view(-1, 576), qkv.view(-1, 576), w, None if w_f32 else w16, xd, fin)
        dxb2 = _ops.qkv_dgrad_finish(dqkv.clone().view(-1, 576), qkv.view(-1, 576), w, None if w_f32 else w16, xd, fin)
    finally:
        _ops.DGRAD_RS_MIN_ROWS = old
    assert torch.equal(db[:, :, 2], dqkv[:, :, 2])                       # dv untouched
    assert not torch.equal(db[:, :, 1], dqkv[:, :, 1])                   # dk took the pooling term
    assert torch.equal(da, db), float((da.float() - db.float()).abs().max())
    assert torch.equal(dxa, dxb) and torch.equal(dxb, dxb2)
    # and against the definition in fp64 (the t correction and both pooling terms)
    ref = dqkv.double().clone()
    cell = (torch.arange(N, device="cuda") // W // r) * (W // r) + (torch.arange(N, device="cuda") % W) // r
    ref[:, :, 0] += dpq.view(B, h, L, d).double()[:, :, cell].permute(0, 2, 1, 3) / (r * r)
    ref[:, :, 1] += dpk.view(B, h, L, d).double()[:, :, cell].permute(0, 2, 1, 3) / (r * r)
    if has_t:
        t = torch.exp(scale * torch.einsum("bhnd,bhcd->bhcn", q.double(), qbar.view(B, h, C, d).double()) - lse_t.view(B, h, C, 1).double())
        ref[:, :, 0] -= scale * torch.einsum("bhcn,bhcd->bnhd", t, uq.view(B, h, C, d).double())
    err = float((db.double() - ref).abs().max() / ref.abs().max())
    assert err < (2e-2 if dtype == "bf16" else 3e-3), err


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("rows,K,NO,a_f32,y_f32,transposed", [
    (100352, 192, 576, 1, 0, 0), (100352, 192, 192, 0, 0, 0), (100352, 192, 192, 0, 1, 1), (1001, 64, 128, 1, 1, 0),
    (50017, 128, 384, 0, 0, 1), (30001, 256, 256, 1, 1, 1), (77, 192, 576, 1, 0, 0), (4097, 256, 512, 0, 0, 1)])
def test_linear_w32_equals_cast_weight(rows, K, NO, a_f32, y_f32, transposed, dtype):
    """ea_linear_w32 (the product straight from the fp32 master weight, optionally read as the transposed operand) is
    BIT-identical to ea_linear on `weight.to(dtype)` / `weight.t().contiguous().to(dtype)`: the rounding of the weight
    happens while it is staged and is the same round-to-nearest-even."""
    import torch
    from efficient_attention import _ops
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    g = torch.Generator(device="cuda").manual_seed(rows + K + NO + transposed)
    a = torch.randn(rows, K, device="cuda", generator=g)
    a = a if a_f32 else a.to(td)
    w32 = torch.randn((K, NO) if transposed else (NO, K), device="cuda", generator=g) * K ** -0.5
    b = None if transposed else torch.randn(NO, device="cuda", generator=g)
    assert _ops.ea_linear_w32_supported(a, w32, td, transposed=bool(transposed))
    od = torch.float32 if y_f32 else td
    y, ac = _ops.ea_linear(a, w32, b, od, want_cast=bool(a_f32), elem_dtype=td, transposed=bool(transposed))
    w16 = (w32.t().contiguous() if transposed else w32).to(td)
    y0, ac0 = _ops.ea_linear(a, w16, b, od, want_cast=bool(a_f32))
    assert torch.equal(y, y0)
    if a_f32:
        assert torch.equal(ac, ac0)


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,r", [(128, 28, 28, 4), (3, 28, 28, 4), (5, 14, 14, 2), (128, 14, 14, 2), (2, 8, 12, 4), (1, 4, 4, 4)])
@pytest.mark.parametrize("a_f32", [1, 0])
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_linear_pool_equals_projection_plus_chunk_mean(B, H, W, r, a_f32, dtype, monkeypatch):
    """ea_linear_w32_pool (round 4): the 192 -> 576 projection that walks the tokens cell by cell and emits the r x r
    pooled q / k rows from its epilogue.  qkv and the rounded copy of x: BIT-identical to ea_linear_w32; pooled rows: the
    means ea_eva_chunk_mean_fwd computes from the stored rows (same rounded values, another summation order)."""
    import ctypes
    import torch
    from efficient_attention import _ops
    from efficient_attention import _native as nv
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + H * 10 + r + a_f32)
    rows, h, d = B * H * W, 3, 64
    x = torch.randn(rows, 192, device="cuda", generator=g)
    x = x if a_f32 else x.to(td)
    w32 = torch.randn(576, 192, device="cuda", generator=g) * 192 ** -0.5
    b = torch.randn(576, device="cuda", generator=g)
    monkeypatch.setattr(_ops, "USE_PROJ_POOL", True)          # (the kernel under test, whatever EA_PROJ_POOL says)
    assert _ops.proj_pool_supported(x, w32, td, B, H, W, r, 3)
    L = (H // r) * (W // r)
    pq = torch.full((B * h, L, d), float("nan"), device="cuda")
    pk = torch.full((B * h, L, d), float("nan"), device="cuda")
    y, xc = _ops.project_qkv_pooled(x, w32, b, td, bool(a_f32), B, H, W, r, pq, pk)
    y0, xc0 = _ops.ea_linear(x, w32, b, td, want_cast=bool(a_f32), elem_dtype=td)
    assert torch.equal(y, y0)
    if a_f32:
        assert torch.equal(xc, xc0)
    qkv5 = y0.view(B, H * W, 3, h, d)
    q, k, _ = _ops._qkv_views(qkv5)
    pgeom = nv.make_geom(B, h, H * W, d, nv.io_dtype(qkv5), True, (H, W), r, 0, r, L)
    pq0 = torch.empty_like(pq)
    pk0 = torch.empty_like(pk)
    tq, tk = nv.t4(q), nv.t4(k)
    nv.call("ea_eva_chunk_mean_fwd", ctypes.byref(pgeom), ctypes.byref(tq), ctypes.byref(tk), None, nv.ptr(pq0), nv.ptr(pk0),
            nv.stream())
    assert torch.isfinite(pq).all() and torch.isfinite(pk).all()
    # (1) what the kernel forms: W (mean of the rounded x rows of the cell) + b with 16-bit W, b -- in fp64
    xr = (x.to(td) if a_f32 else x).double().view(B, H // r, r, W // r, r, 192).mean((2, 4)).reshape(B * L, 192)
    exact = xr @ w32.to(td).double().t() + b.to(td).double()
    exact = exact.view(B, L, 3, h, d).permute(2, 0, 3, 1, 4).reshape(3, B * h, L, d)
    scale = float(exact.abs().max())
    for got, t in ((pq, exact[0]), (pk, exact[1])):
        assert (got.double() - t).abs().max() <= 2e-5 * scale
    # (2) the cell means of the STORED (rounded) q / k rows -- ea_eva_chunk_mean_fwd, and plain torch -- differ from that by
    # the rows' final rounding averaged over the cell: a fraction of one 16-bit ulp of the largest value
    ulp = 2.0 ** (-8 if dtype == "bf16" else -11)
    ref = y0.float().view(B, H // r, r, W // r, r, 3, h, d).mean((2, 4)).permute(3, 0, 4, 1, 2, 5).reshape(3, B * h, L, d)
    for got, other, t in ((pq, pq0, ref[0]), (pk, pk0, ref[1])):
        assert (other - t).abs().max() <= 1e-5 * scale
        assert (got - t).abs().max() <= 0.75 * ulp * scale


@pytest.mark.gpu
def test_linear_fn_master_weight_path_and_frozen_weight():
    """LinearFn under autocast with fp32 parameters: forward and the output projection's input gradient run from the
    master weight (no cast / transpose kernels), results bit-identical to the cast-weight path; a frozen weight with a
    trainable bias and an fp32 input (`xl is None`) differentiates."""
    import torch
    from efficient_attention import _ops
    torch.manual_seed(5)
    lin = torch.nn.Linear(192, 192).cuda()
    x = torch.randn(4, 50, 192, device="cuda", requires_grad=True)
    gy = torch.randn(4, 50, 192, device="cuda")
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = _ops.linear(x, lin)
    y.backward(gy.to(y.dtype))
    dx, dw, db = x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone()
    # reference: the same kernels on explicitly cast operands
    w16 = lin.weight.detach().to(torch.bfloat16)
    y0, xc = _ops.ea_linear(x.detach().reshape(-1, 192), w16, lin.bias.detach().float(), torch.bfloat16, want_cast=True)
    assert torch.equal(y.reshape(-1, 192), y0)
    dx0, _ = _ops.ea_linear(gy.to(torch.bfloat16).reshape(-1, 192), w16.t().contiguous(), None, torch.float32)
    assert torch.equal(dx.reshape(-1, 192), dx0)
    dw0, db0 = _ops.wgrad(gy.to(torch.bfloat16).reshape(-1, 192), xc, True)
    assert torch.equal(dw, dw0) and torch.equal(db, db0)
    # frozen weight, trainable bias, fp32 input
    lin.weight.requires_grad_(False)
    lin.bias.grad = None
    lin.weight.grad = None
    x2 = torch.randn(4, 50, 192, device="cuda", requires_grad=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y2 = _ops.linear(x2, lin)
    y2.backward(gy.to(y2.dtype))
    assert x2.grad is not None and lin.bias.grad is not None and lin.weight.grad is None
    assert torch.allclose(lin.bias.grad, gy.to(torch.bfloat16).float().sum((0, 1)), rtol=1e-3, atol=1e-3)


@pytest.mark.gpu
def test_colsum2_matches_two_colsums():
    import torch
    from efficient_attention import _ops
    g = torch.Generator(device="cuda").manual_seed(3)
    a = torch.randn(384, 8192, device="cuda", generator=g)
    b = torch.randn(384, 384, device="cuda", generator=g)
    sa, sb = _ops.colsum2_f32(a, b)
    assert torch.equal(sa, _ops.colsum_f32(a)) and torch.equal(sb, _ops.colsum_f32(b))
    assert torch.allclose(sa.double(), a.double().sum(0), atol=1e-4)


@pytest.mark.gpu
def test_linear_unsupported_geometry_falls_to_library():
    """Geometries the streaming kernel is not built for are refused by the C ABI (EA_E_UNSUPPORTED), and the autograd
    Function routes them to the library GEMM."""
    import torch
    from efficient_attention import _ops
    assert _ops.nv.lib().ea_linear_supported(576, 192) == 0 and not _ops._lin_geometry(576, 192)
    assert _ops.nv.lib().ea_linear_supported(192, 576) != 0 and _ops._lin_geometry(192, 576)
    for K in (64, 128, 192, 256, 320, 512):
        for NO in (64, 96, 128, 192, 320, 384, 576, 768, 1024, 1536):
            assert (_ops.nv.lib().ea_linear_supported(K, NO) != 0) == _ops._lin_geometry(K, NO), (K, NO)
    x = torch.randn(300, 320, device="cuda", requires_grad=True)
    lin = torch.nn.Linear(320, 960).cuda()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = _ops.linear(x, lin)
    y.float().sum().backward()
    assert x.grad is not None and lin.weight.grad is not None


@pytest.mark.gpu
@pytest.mark.parametrize("mis_type", ["mis-opt", "mis-biased", "mis-bh"])
@pytest.mark.parametrize("mode,L,d", [(1, 49, 64), (2, 49, 64), (0, 100, 64), (1, 16, 32), (2, 64, 128)])
def test_lara_sample_matches_torch(mode, L, d, mis_type):
    """ea_lara_sample_fwd/bwd (sampling + [C x L] proposal densities for C > 64) against the same algebra written with
    torch ops in fp64, values and gradients."""
    import torch
    from efficient_attention import _ops
    g = torch.Generator(device="cuda").manual_seed(17 * L + d + mode)
    B, h = 2, 3
    C = L * (2 if mode else 1)
    q_bar = torch.randn(B, h, L, d, device="cuda", generator=g).requires_grad_(True)
    mu = (torch.randn(B, h, L, d, device="cuda", generator=g) * 0.7).requires_grad_(True)
    noise = None if mode == 0 else torch.randn(B, h, L if mode == 1 else C, d, device="cuda", generator=g)
    scale = d ** -0.5
    mis = _ops.MIS[mis_type]
    assert _ops._sample_fits(L, C, d)
    outs = _ops.LaraSampleFn.apply(q_bar, mu, noise, mis, mode, scale)
    gs = [torch.randn(o.shape, device="cuda", generator=g) if o is not None else None for o in outs]
    loss = sum((o * w).sum() for o, w in zip(outs, gs) if o is not None)
    dq, dm = torch.autograd.grad(loss, [q_bar, mu], allow_unused=True)

    q64, m64 = q_bar.detach().double().requires_grad_(True), mu.detach().double().requires_grad_(True)
    n64 = None if noise is None else noise.double()
    if mode == 0:
        om = m64
    elif mode == 2:
        om = m64.repeat(1, 1, 2, 1) + n64
    else:
        om = torch.cat([m64 + n64, m64 - n64], dim=-2)
    rep = (lambda t: t.repeat(1, 1, 2, 1)) if mode else (lambda t: t)

    def prm(data, proj):
        return scale * torch.einsum("bhcd,bhnd->bhcn", proj, data) - 0.5 * scale * (data * data).sum(-1).unsqueeze(-2)
    qr = bh = None
    if mis == 0:
        lpmu = prm(rep(m64), om)
        lp = torch.diagonal(lpmu, dim1=-1, dim2=-2)
        bh = torch.exp(lp - torch.logsumexp(lpmu, dim=-1))
        qr = rep(q64)
    else:
        lp = torch.logsumexp(prm(m64, om), dim=-1)
        qr = rep(m64) if mis == 1 else None
    refs = (om, qr, bh, lp)
    for o, r in zip(outs, refs):
        assert (o is None) == (r is None)
        if o is not None:
            assert torch.allclose(o.double(), r, rtol=2e-4, atol=2e-5), float((o.double() - r).abs().max())
    loss64 = sum((r * w.double()).sum() for r, w in zip(refs, gs) if r is not None)
    rq, rm = torch.autograd.grad(loss64, [q64, m64], allow_unused=True)
    rq = torch.zeros_like(q64) if rq is None else rq
    assert torch.allclose(dm.double(), rm, rtol=1e-3, atol=1e-3 * float(rm.abs().max())), float((dm.double() - rm).abs().max())
    assert torch.allclose(dq.double(), rq, rtol=1e-3, atol=1e-4 + 1e-3 * float(rq.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("mis,mixed,has_mlp,dup", [(0, 1, 1, 0), (1, 0, 1, 0), (2, 1, 0, 0), (0, 0, 1, 1)])
def test_lara_composite_equals_step_by_step(mis, mixed, has_mlp, dup):
    """ea_lara_layer_fwd / _bwd (the whole LARA core as one C-ABI call each way, on caller-owned workspaces) against the
    step-by-step launch sequence they replace: same kernels, same order -> bit-identical outputs and gradients."""
    import os
    import torch
    from efficient_attention import _ops
    torch.manual_seed(mis * 7 + mixed)
    B, H, W, h, d, r = 4, 28, 28, 3, 64, 4
    L = (H // r) * (W // r)
    if dup:
        B, H, W, r = 2, 16, 16, 4                       # 16 landmarks x 2 antithetic samples
        L = 16
    C = L * (2 if dup else 1)
    qkv = (0.5 * torch.randn(B, H * W, 3, h, d, device="cuda")).bfloat16()
    noise = torch.randn(B, h, C // (2 if dup == 1 else 1) if dup == 1 else C, d, device="cuda")
    dout = torch.randn(B, H * W, h, d, device="cuda").bfloat16()
    params = []
    if has_mlp:
        for _ in range(2):
            params += [torch.randn(d, d, device="cuda") * d ** -0.5, torch.randn(d, device="cuda") * 0.1,
                       1 + 0.1 * torch.randn(d, device="cuda"), 0.1 * torch.randn(d, device="cuda")]
    icfg = [H, W, r, has_mlp, mixed, mis, dup, 1]
    fcfg = [2.0, d ** -0.5]
    res = {}
    # "1": composite, merge launches folded into their consumers (round 5, the default); "1u": composite with the merge
    # launches (EA_LARA_FOLD=0); "0": step by step
    for mode in ("1", "1u", "0"):
        os.environ["EA_LARA_COMPOSITE"] = mode[0]
        if mode == "1u":
            os.environ["EA_LARA_FOLD"] = "0"
        try:
            outs = torch.ops.ea.lara_fwd(qkv, None, noise, icfg, fcfg, params)
            grads = torch.ops.ea.lara_bwd(dout, qkv, None, noise, list(outs[1:]), icfg, fcfg, params)
        finally:
            os.environ.pop("EA_LARA_COMPOSITE", None)
            os.environ.pop("EA_LARA_FOLD", None)
        res[mode] = (len(outs), outs[0], grads)
    assert res["1"][0] == 2 and res["1u"][0] == 2 and res["0"][0] > 2          # one workspace vs the individual tensors
    assert torch.equal(res["1u"][1], res["0"][1])
    assert len(res["1u"][2]) == len(res["0"][2]) == len(res["1"][2])
    for a, b in zip(res["1u"][2], res["0"][2]):
        assert torch.equal(a, b)
    # folded: the forward merge repeats the merge kernel's arithmetic (bit-identical output); the backward's dkk = dkv . kv is
    # summed in another order (8 channels per lane instead of 4): equal to fp32 rounding
    assert torch.equal(res["1"][1], res["0"][1])
    for a, b in zip(res["1"][2], res["0"][2]):
        tol = 2e-2 if a.dtype in (torch.bfloat16, torch.float16) else 2e-4
        assert float((a.float() - b.float()).abs().max()) <= tol * max(float(b.float().abs().max()), 1e-6), (a.dtype, a.shape)


@pytest.mark.gpu
@pytest.mark.parametrize("gen,train", [("pool-mixed", True), ("pool", True), ("pool-mixed", False)])
def test_lara_module_single_node_equals_three_nodes(gen, train):
    """LinearRA's common 2-D training case runs as ONE autograd node (_ops.LaraModuleFn: qkv projection + core + output
    projection); it issues the same kernels in the same order as the three nodes it replaces -> bit-identical output,
    input gradient and parameter gradients."""
    import warnings
    import torch
    import efficient_attention as ea
    from efficient_attention import _ops
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.manual_seed(11)
        m = ea.AttentionFactory.build_attention("lara", dict(dim=192, num_heads=3, num_landmarks=49, proposal_gen=gen,
                                                             mis_type="mis-opt", alpha_coeff=2.0)).cuda()
    m.train(train)
    x0 = torch.randn(4, 28, 28, 192, device="cuda")
    g = torch.randn(4, 28, 28, 192, device="cuda").bfloat16()
    res = {}
    # (the single node's projection also emits the pooled q / k rows -- round 4, ea_linear_w32_pool: cell means of the
    #  UNROUNDED product instead of the stored rows; switched off here for the bit-for-bit comparison and compared on its
    #  own in test_lara_module_pooled_projection_close_to_separate_pooling)
    old_pool, old_ms = _ops.USE_PROJ_POOL, _ops.USE_MULTI_SUM
    _ops.USE_PROJ_POOL = False
    _ops.USE_MULTI_SUM = False        # (one launch for all terminal sums: another order of additions for the landmark parameters)
    for single in (True, False):
        old = _ops.USE_LARA_MODULE_FN
        _ops.USE_LARA_MODULE_FN = single
        try:
            for p in m.parameters():
                p.grad = None
            x = x0.clone().requires_grad_(True)
            torch.manual_seed(5)                          # the same landmark noise in both runs
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m(x)
            node = type(y.grad_fn).__name__
            y.backward(g)
            res[single] = (node, y.detach(), x.grad, {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
        finally:
            _ops.USE_LARA_MODULE_FN = old
    _ops.USE_PROJ_POOL, _ops.USE_MULTI_SUM = old_pool, old_ms
    assert res[True][0].startswith("LaraModuleFn") and not res[False][0].startswith("LaraModuleFn")
    assert torch.equal(res[True][1], res[False][1])
    assert torch.equal(res[True][2], res[False][2])
    assert res[True][3].keys() == res[False][3].keys() and len(res[True][3]) >= 4
    for n in res[True][3]:
        assert torch.equal(res[True][3][n], res[False][3][n]), n


@pytest.mark.gpu
def test_multi_sum_equals_fp64_sums_and_single_reductions():
    """ea_multi_sum: up to six slice reductions in one launch -- equal to fp64 sums to fp32 accuracy, reproducible, and
    bit-identical to ea_part_sum on the segments of up to 96 slices (the many-slice segments -- the per-(b,h) partials of
    the landmark parameters -- take a wider slice-lane layout with its own fixed order of additions, round 5)."""
    import torch
    from efficient_attention import _ops
    from efficient_attention import _native as nv
    g = torch.Generator(device="cuda").manual_seed(3)
    shapes = [(12, 37056), (12, 111168), (384, 8192), (384, 384), (512, 9408), (3, 4)]
    parts = [torch.randn(S, n, device="cuda", generator=g) for S, n in shapes]
    outs = _ops.multi_sum(parts)
    outs2 = _ops.multi_sum(parts)
    for p, o, o2 in zip(parts, outs, outs2):
        assert torch.equal(o, o2)
        ref = p.double().sum(0)
        assert float((o.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) * max(1.0, p.shape[0] ** 0.5 / 4)
        single = torch.empty_like(o)
        nv.call("ea_part_sum", p.shape[0], p.shape[1], p.shape[1], nv.ptr(p), nv.ptr(single), nv.stream())
        if p.shape[0] <= 96:
            assert torch.equal(o, single)
        else:
            assert float((o - single).abs().max()) <= 1e-5 * float(ref.abs().max())
    one = _ops.multi_sum(parts[:1])
    assert torch.equal(one[0], outs[0])


@pytest.mark.gpu
def test_lara_module_deferred_sums_equal_separate_reductions(monkeypatch):
    """LaraModuleFn's backward with the terminal sums in one launch (EA_MULTI_SUM) against the separate reductions: the
    projection gradients bit for bit (same order of additions), the landmark parameters to fp32 accuracy."""
    import warnings
    import torch
    import efficient_attention as ea
    from efficient_attention import _ops
    monkeypatch.setattr(_ops, "USE_WGRAD_PAIR", False)      # (the paired launch cuts the token axis into other slices)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.manual_seed(15)
        m = ea.AttentionFactory.build_attention("lara", dict(dim=192, num_heads=3, num_landmarks=49, proposal_gen="pool-mixed",
                                                             mis_type="mis-opt", alpha_coeff=2.0)).cuda()
    m.train()
    x0 = torch.randn(4, 28, 28, 192, device="cuda")
    g = torch.randn(4, 28, 28, 192, device="cuda").bfloat16()
    res = {}
    for ms in (True, False):
        old = _ops.USE_MULTI_SUM
        _ops.USE_MULTI_SUM = ms
        try:
            for p in m.parameters():
                p.grad = None
            x = x0.clone().requires_grad_(True)
            torch.manual_seed(5)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m(x)
            y.backward(g)
            res[ms] = (x.grad, {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
        finally:
            _ops.USE_MULTI_SUM = old
    assert torch.equal(res[True][0], res[False][0])
    assert res[True][1].keys() == res[False][1].keys()
    for n in res[True][1]:
        a, b = res[True][1][n], res[False][1][n]
        if n.startswith(("qkv.", "proj.")):
            assert torch.equal(a, b), n
        else:
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-9, n


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("composite", ["1", "0"])
def test_lara_module_pooled_projection_close_to_separate_pooling(dtype, composite, monkeypatch):
    """The projection-with-pooling path of the single node (ea_linear_w32_pool -> ea_lara_layer_fwd with
    EA_LARA_POOLED_READY, or the step-by-step launches with the pooled rows handed over) against the separate pooling
    pass: the pooled rows differ by a fraction of a 16-bit ulp (unrounded vs rounded rows), so outputs and gradients agree
    far inside the module tolerances."""
    import warnings
    import torch
    import efficient_attention as ea
    from efficient_attention import _ops
    monkeypatch.setenv("EA_LARA_COMPOSITE", composite)
    monkeypatch.setattr(_ops, "USE_LARA_MODULE_FN", True)
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.manual_seed(13)
        m = ea.AttentionFactory.build_attention("lara", dict(dim=192, num_heads=3, num_landmarks=49, proposal_gen="pool-mixed",
                                                             mis_type="mis-opt", alpha_coeff=2.0)).cuda()
    m.train()
    x0 = torch.randn(4, 28, 28, 192, device="cuda")
    g = torch.randn(4, 28, 28, 192, device="cuda").to(td)
    res = {}
    for pool in (True, False):
        old = _ops.USE_PROJ_POOL
        _ops.USE_PROJ_POOL = pool
        try:
            for p in m.parameters():
                p.grad = None
            x = x0.clone().requires_grad_(True)
            torch.manual_seed(5)
            with torch.autocast("cuda", dtype=td):
                y = m(x)
            assert type(y.grad_fn).__name__.startswith("LaraModuleFn")
            y.backward(g)
            res[pool] = (y.detach().float(), x.grad, {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
        finally:
            _ops.USE_PROJ_POOL = old
    tol = 1.6e-2 if dtype == "bf16" else 2e-3           # two ulps of the 16-bit outputs at the largest value

    def close(a, b, what):
        sc = float(b.abs().max())
        assert float((a - b).abs().max()) <= tol * sc, (what, float((a - b).abs().max()) / sc)
    close(res[True][0], res[False][0], "y")
    close(res[True][1], res[False][1], "dx")
    assert res[True][2].keys() == res[False][2].keys()
    tol *= 4                      # parameter gradients: sums over all tokens with cancellation (LayerNorm gains most of all)
    for n in res[True][2]:
        close(res[True][2][n], res[False][2][n], n)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("grid,window,landmarks", [((28, 28), 7, 49), ((14, 14), 7, 49)])
def test_eva_pooled_projection_close_to_separate_chunk_means(dtype, grid, window, landmarks):
    """EVA on a 2-D grid without window extension: the chunk means of q, k come out of the qkv projection
    (_ops.LinearPoolFn -> ea_linear_w32_pool) instead of ea_eva_chunk_mean_fwd; outputs and every gradient agree with the
    separate pass far inside the module tolerances, and the hint path is really taken."""
    import warnings
    import torch
    import efficient_attention as ea
    from efficient_attention import _ops
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.manual_seed(14)
        m = ea.AttentionFactory.build_attention("eva", dict(dim=192, num_heads=3, num_landmarks=landmarks, window_size=window,
                                                            attn_2d=True, use_rpe=True, adaptive_proj="default")).cuda()
    m.train()
    x0 = torch.randn(3, grid[0], grid[1], 192, device="cuda")
    g = torch.randn(3, grid[0], grid[1], 192, device="cuda").to(td)
    res = {}
    for pool in (True, False):
        old = _ops.USE_PROJ_POOL
        _ops.USE_PROJ_POOL = pool
        calls = []
        orig = _ops.project_qkv_pooled
        orig_w = _ops.project_qkv_wsw                  # (round 6: the prepared-weight flavour of the same kernel)
        _ops.project_qkv_pooled = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        _ops.project_qkv_wsw = lambda *a, **k: (calls.append(1) if (len(a) > 6 and a[6] is not None) or k.get("grid") is not None
                                                else None, orig_w(*a, **k))[1]
        try:
            for p in m.parameters():
                p.grad = None
            x = x0.clone().requires_grad_(True)
            torch.manual_seed(5)
            with torch.autocast("cuda", dtype=td):
                y = m(x)
            y.backward(g)
            res[pool] = (y.detach().float(), x.grad, {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
        finally:
            _ops.USE_PROJ_POOL = old
            _ops.project_qkv_pooled = orig
            _ops.project_qkv_wsw = orig_w
        assert len(calls) == (1 if pool else 0)
    tol = 1.6e-2 if dtype == "bf16" else 2e-3           # two ulps of the 16-bit outputs at the largest value

    def close(a, b, what):
        sc = float(b.abs().max())
        assert float((a - b).abs().max()) <= tol * sc, (what, float((a - b).abs().max()) / sc)
    close(res[True][0], res[False][0], "y")
    close(res[True][1], res[False][1], "dx")
    assert res[True][2].keys() == res[False][2].keys()
    tol *= 4                      # parameter gradients: sums over all tokens with cancellation (LayerNorm gains most of all)
    for n in res[True][2]:
        close(res[True][2][n], res[False][2][n], n)


@pytest.mark.gpu
@pytest.mark.parametrize("frozen", ["qkv.weight", "proj.weight", "both"])
def test_lara_module_single_node_bias_only_finetuning(frozen, monkeypatch):
    """frozen projection weight + trainable bias + fp32 x under autocast (bias-only fine-tuning).  The single
    node must not ask for the rounded input it did not keep: the bias gradient is a column sum of d qkv (ea_bias_grad).
    Same gradients as the three-node path, bit for bit."""
    import warnings
    import torch
    import efficient_attention as ea
    from efficient_attention import _ops
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.manual_seed(12)
        m = ea.AttentionFactory.build_attention("lara", dict(dim=192, num_heads=3, num_landmarks=49, proposal_gen="pool-mixed",
                                                             mis_type="mis-opt", alpha_coeff=2.0, qkv_bias=True)).cuda()
    m.train()
    for n, p in m.named_parameters():
        if n == frozen or (frozen == "both" and n in ("qkv.weight", "proj.weight")):
            p.requires_grad_(False)
    assert m.qkv.bias is not None and m.qkv.bias.requires_grad
    x0 = torch.randn(2, 28, 28, 192, device="cuda")
    g = torch.randn(2, 28, 28, 192, device="cuda").bfloat16()
    res = {}
    monkeypatch.setattr(_ops, "USE_PROJ_POOL", False)      # (pooled rows from the projection: not bit-equal to the three nodes)
    for single in (True, False):
        old = _ops.USE_LARA_MODULE_FN
        _ops.USE_LARA_MODULE_FN = single
        try:
            for p in m.parameters():
                p.grad = None
            x = x0.clone().requires_grad_(True)
            torch.manual_seed(5)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m(x)
            node = type(y.grad_fn).__name__
            y.backward(g)
            res[single] = (node, x.grad, {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
        finally:
            _ops.USE_LARA_MODULE_FN = old
    assert res[True][0].startswith("LaraModuleFn")
    assert torch.equal(res[True][1], res[False][1])
    assert res[True][2].keys() == res[False][2].keys() and "qkv.bias" in res[True][2] and "proj.bias" in res[True][2]
    for n in res[True][2]:
        a, b = res[True][2][n], res[False][2][n]
        # the one-pass weight gradient carries the bias sum along; alone it is ea_bias_grad: same sum, other order
        assert torch.allclose(a, b, rtol=2e-3, atol=2e-3 * float(b.abs().max())), n


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("masked,bias", [(False, True), (True, True), (False, False)])
def test_lara_adaptive_1d_fold_kernels_match_framework_fold(dtype, masked, bias):
    """LARA 'adaptive-1d' with the generators' Linear folded into the qkv projection: extended weight built and
    differentiated by ea_lara_fold_fwd / _bwd (round 4) against the framework-op construction it replaces -- y, dx and every
    parameter gradient (the fp32 fold differs in summation order only; its 16-bit rounding can flip single weights by one
    ulp)."""
    import warnings
    import torch
    import efficient_attention as ea
    from efficient_attention import _ops
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.manual_seed(21)
        m = ea.AttentionFactory.build_attention("lara", dict(dim=512, num_heads=8, num_landmarks=16, proposal_gen="adaptive-1d",
                                                             mis_type="mis-opt", qkv_bias=bias)).cuda()
    m.train()
    B, N = 2, 1000
    x0 = torch.randn(B, N, 512, device="cuda")
    g = torch.randn(B, N, 512, device="cuda").to(td)
    mask = None
    if masked:
        mask = torch.zeros(B, N, dtype=torch.bool, device="cuda")
        mask[0, 900:] = True
    res = {}
    for fold in (True, False):
        old = _ops.USE_FOLD_KERNELS
        _ops.USE_FOLD_KERNELS = fold
        try:
            for p in m.parameters():
                p.grad = None
            x = x0.clone().requires_grad_(True)
            torch.manual_seed(5)
            with torch.autocast("cuda", dtype=td):
                y = m(x, mask)
            y.backward(g)
            res[fold] = (y.detach().float(), x.grad, {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
        finally:
            _ops.USE_FOLD_KERNELS = old
    tol = 1.6e-2 if dtype == "bf16" else 2e-3

    def close(a, b, what, t):
        sc = float(b.abs().max())
        assert float((a - b).abs().max()) <= t * sc, (what, float((a - b).abs().max()) / sc)
    close(res[True][0], res[False][0], "y", tol)
    close(res[True][1], res[False][1], "dx", tol)
    assert res[True][2].keys() == res[False][2].keys() and len(res[True][2]) >= 8
    for n in res[True][2]:
        close(res[True][2][n], res[False][2][n], n, 4 * tol)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("B,h,N,L", [(2, 8, 4096, 49), (1, 2, 1000, 16), (2, 3, 333, 7), (1, 1, 130, 64),
                                     (8, 8, 2048, 49),      # several segments per wave: the row queue crosses segment boundaries
                                     (1, 1, 17, 16)])       # one- and two-token segments
def test_lara_seglin_matches_fp64_torch(dtype, B, h, N, L):
    """ea_lara_seglin_fwd / _bwd (round 4): segment means of LayerNorm(G x + b) straight from the stored q / k rows, the
    generator Linear on the MFMA inside the kernel -- against fp64 torch on the same 16-bit-rounded rows and generator
    weights: means, the gradient accumulated into dq / dk (on top of a pre-filled buffer), dG, d g_b, d ln_w, d ln_b."""
    import torch
    from efficient_attention import _ops
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    gen = torch.Generator(device="cuda").manual_seed(N + L)
    qkv = (torch.randn(B, N, 3, h, 64, device="cuda", generator=gen)).to(td)
    ps = [torch.randn(64, 64, device="cuda", generator=gen) * 0.15, torch.randn(64, device="cuda", generator=gen) * 0.1,
          torch.randn(64, 64, device="cuda", generator=gen) * 0.15, torch.randn(64, device="cuda", generator=gen) * 0.1,
          1 + 0.2 * torch.randn(64, device="cuda", generator=gen), 0.1 * torch.randn(64, device="cuda", generator=gen),
          1 + 0.2 * torch.randn(64, device="cuda", generator=gen), 0.1 * torch.randn(64, device="cuda", generator=gen)]
    ps = [p_.requires_grad_(True) for p_ in ps]
    slot = _ops._GradSlot()
    base = (torch.randn(B, N, 3, h, 64, device="cuda", generator=gen) * 0.01).to(td)
    slot.buf = base.clone()
    qbar, kbar = _ops.SegLinLnMeanFn.apply(qkv, L, slot, *ps)
    gq = torch.randn(B, h, L, 64, device="cuda", generator=gen)
    gk = torch.randn(B, h, L, 64, device="cuda", generator=gen)
    buf = slot.buf
    (qbar * gq).sum().backward(retain_graph=True)
    # (one backward call handles both sides: run it once with both cotangents)
    for p_ in ps:
        p_.grad = None
    slot.buf = base.clone()
    buf = slot.buf
    torch.autograd.backward([qbar, kbar], [gq, gk])
    # fp64 reference on the rounded values
    x = qkv.double().cpu().requires_grad_(True)
    rp = [p_.detach().to(td).double().cpu().requires_grad_(True) if i in (0, 2) else p_.detach().double().cpu().requires_grad_(True)
          for i, p_ in enumerate(ps)]
    segs = N // L
    nshort = L if N % L == 0 else (segs + 1) * L - N

    def side(xs, G, gb, lw, lb):
        z = torch.einsum("bnhd,ed->bnhe", xs, G) + gb
        y = torch.nn.functional.layer_norm(z, (64,), lw, lb, 1e-5).permute(0, 2, 1, 3)          # [B,h,N,64]
        head = y[:, :, :nshort * segs].reshape(B, h, nshort, segs, 64).mean(-2)
        if nshort == L:
            return head
        tail = y[:, :, nshort * segs:].reshape(B, h, L - nshort, segs + 1, 64).mean(-2)
        return torch.cat([head, tail], -2)
    rq = side(x[:, :, 0], rp[0], rp[1], rp[4], rp[5])
    rk = side(x[:, :, 1], rp[2], rp[3], rp[6], rp[7])
    torch.autograd.backward([rq, rk], [gq.double().cpu(), gk.double().cpu()])
    tol = 1.2e-2 if dtype == "bf16" else 2e-3

    def close(a, b, what, t=tol):
        sc = float(b.abs().max())
        assert float((a.double().cpu() - b).abs().max()) <= t * sc, (what, float((a.double().cpu() - b).abs().max()) / sc)
    close(qbar, rq.detach(), "qbar", tol / 4)
    close(kbar, rk.detach(), "kbar", tol / 4)
    dgrad = (buf.double().cpu() - base.double().cpu())
    close(dgrad[:, :, 0], x.grad[:, :, 0], "dq", 2 * tol)           # (accumulated in the 16-bit buffer: its own rounding)
    close(dgrad[:, :, 1], x.grad[:, :, 1], "dk", 2 * tol)
    assert float(dgrad[:, :, 2].abs().max()) == 0.0
    for i, nm in enumerate(["Gq", "gqb", "Gk", "gkb", "lqw", "lqb", "lkw", "lkb"]):
        close(ps[i].grad, rp[i].grad, nm)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("masked", [False, True])
def test_lara_adaptive_1d_seglin_matches_folded_path(dtype, masked):
    """The module with the generator inside the segment kernels (round 4 default) against the folded-projection path it
    replaces (EA_SEGLIN=0): y, dx and every parameter gradient."""
    import warnings
    import torch
    import efficient_attention as ea
    from efficient_attention import _ops
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.manual_seed(23)
        m = ea.AttentionFactory.build_attention("lara", dict(dim=512, num_heads=8, num_landmarks=16, proposal_gen="adaptive-1d",
                                                             mis_type="mis-opt")).cuda()
    m.train()
    B, N = 2, 1000
    x0 = torch.randn(B, N, 512, device="cuda")
    g = torch.randn(B, N, 512, device="cuda").to(td)
    mask = None
    if masked:
        mask = torch.zeros(B, N, dtype=torch.bool, device="cuda")
        mask[0, 900:] = True
    res = {}
    for sl in (True, False):
        old = _ops.USE_SEGLIN
        _ops.USE_SEGLIN = sl
        try:
            for p in m.parameters():
                p.grad = None
            x = x0.clone().requires_grad_(True)
            torch.manual_seed(5)
            with torch.autocast("cuda", dtype=td):
                y = m(x, mask)
            y.backward(g)
            res[sl] = (y.detach().float(), x.grad, {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
        finally:
            _ops.USE_SEGLIN = old
    tol = 2.4e-2 if dtype == "bf16" else 3e-3

    def close(a, b, what, t):
        sc = float(b.abs().max())
        assert float((a - b).abs().max()) <= t * sc, (what, float((a - b).abs().max()) / sc)
    close(res[True][0], res[False][0], "y", tol)
    close(res[True][1], res[False][1], "dx", tol)
    assert res[True][2].keys() == res[False][2].keys() and len(res[True][2]) >= 8
    for n in res[True][2]:
        close(res[True][2][n], res[False][2][n], n, 4 * tol)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("landmarks,masked", [(49, False), (16, False), (16, True), (33, False)])
def test_lara_seglin_fused_finish_matches_separate_pass(dtype, landmarks, masked):
    """ea_lara_seglin_bwd_fin (round 6): the estimator's last dq correction (lara.py:223 differentiated) applied by the segment
    backward's dq / dk pass against the same step with ea_lara_bwd_finish as a pass of its own (EA_SEGLIN_FIN=0).  The two
    differ only in where dq is rounded to the I/O type (once instead of twice): y identical, dx and the parameter gradients
    within the rounding of one more 16-bit store."""
    import warnings
    import torch
    import efficient_attention as ea
    from efficient_attention import _ops
    if not (_ops.USE_SEGLIN and _ops.USE_SEGLIN_FIN):
        pytest.skip("dev switch EA_SEGLIN / EA_SEGLIN_FIN = 0")
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.manual_seed(29)
        m = ea.AttentionFactory.build_attention("lara", dict(dim=512, num_heads=8, num_landmarks=landmarks,
                                                             proposal_gen="adaptive-1d", mis_type="mis-opt")).cuda()
    m.train()
    B, N = 2, 1000
    x0 = torch.randn(B, N, 512, device="cuda")
    g = torch.randn(B, N, 512, device="cuda").to(td)
    mask = None
    if masked:
        mask = torch.zeros(B, N, dtype=torch.bool, device="cuda")
        mask[1, 870:] = True
    calls = {"fin": 0, "sep": 0}
    orig_call = _ops.nv.call

    def counting(name, *a):
        if name == "ea_lara_seglin_bwd_fin":
            calls["fin"] += 1
        if name == "ea_lara_bwd_finish":
            calls["sep"] += 1
        return orig_call(name, *a)
    res = {}
    _ops.nv.call = counting
    try:
        for fused in (True, False):
            old = _ops.USE_SEGLIN_FIN
            _ops.USE_SEGLIN_FIN = fused
            try:
                for p in m.parameters():
                    p.grad = None
                x = x0.clone().requires_grad_(True)
                torch.manual_seed(5)
                with torch.autocast("cuda", dtype=td):
                    y = m(x, mask)
                y.backward(g)
                res[fused] = (y.detach().float(), x.grad, {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
            finally:
                _ops.USE_SEGLIN_FIN = old
    finally:
        _ops.nv.call = orig_call
    assert calls == {"fin": 1, "sep": 1}, calls
    assert torch.equal(res[True][0], res[False][0])
    tol = 8e-3 if dtype == "bf16" else 1e-3

    def close(a, b, what, t):
        sc = float(b.abs().max())
        assert float((a - b).abs().max()) <= t * sc, (what, float((a - b).abs().max()) / sc)
    close(res[True][1], res[False][1], "dx", tol)
    assert res[True][2].keys() == res[False][2].keys() and len(res[True][2]) >= 8
    for n in res[True][2]:
        close(res[True][2][n], res[False][2][n], n, tol)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("dim,heads,grid,window,landmarks,module_fn", [
    (192, 3, (28, 28), 7, 49, True),        # cfg3: single node, chunk means from the projection kernel into the workspace
    (192, 3, (14, 14), 7, 49, True),        # cfg2
    (192, 3, (28, 28), 7, 49, False),       # three nodes: LinearPoolFn hint -> step-by-step in both runs (hint is not in a workspace)
    (128, 2, (48, 48), 8, 36, True),        # PvT stage 2 (cfg4): single node without the pooled projection
    (128, 2, (16, 16), 8, 16, False),       # three nodes, plain projection: EvaAttnFn takes the composite entry
    (64, 2, (16, 16), 4, 16, False),        # d = 32
])
def test_eva_composite_equals_step_by_step(dtype, dim, heads, grid, window, landmarks, module_fn, monkeypatch):
    """ea_eva_layer_fwd / _bwd (one C-ABI call per direction for the 2-D EVA core) issue exactly the launches the
    step-by-step path issues: outputs and every gradient are bit-identical (the bias-table gradient to fp32 summation
    order), with and without the relative-position bias."""
    import warnings
    import torch
    import efficient_attention as ea
    from efficient_attention import _ops
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    monkeypatch.setattr(_ops, "USE_EVA_MODULE_FN", module_fn)
    monkeypatch.setattr(_ops, "USE_LARA_MODULE_FN", True)      # (eva_module_fn_supported builds on lara_module_fn_supported)
    for use_rpe in (True, False):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            torch.manual_seed(15)
            m = ea.AttentionFactory.build_attention("eva", dict(dim=dim, num_heads=heads, num_landmarks=landmarks, window_size=window,
                                                                attn_2d=True, use_rpe=use_rpe, adaptive_proj="default")).cuda()
        m.train()
        with torch.no_grad():
            for p in m.parameters():
                p.add_(0.02 * torch.randn_like(p))
        x0 = torch.randn(3, grid[0], grid[1], dim, device="cuda")
        g = torch.randn(3, grid[0], grid[1], dim, device="cuda").to(td)
        res, used = {}, {}
        for comp in (True, False):
            monkeypatch.setattr(_ops, "_eva_use_composite", lambda comp=comp: comp)
            calls = []
            orig = _ops.nv.call

            def spy(name, *a, _calls=calls, _orig=orig):
                _calls.append(name)
                return _orig(name, *a)
            monkeypatch.setattr(_ops.nv, "call", spy)
            for p in m.parameters():
                p.grad = None
            x = x0.clone().requires_grad_(True)
            torch.manual_seed(5)
            with torch.autocast("cuda", dtype=td):
                y = m(x)
            y.backward(g)
            monkeypatch.setattr(_ops.nv, "call", orig)
            res[comp] = (y.detach().clone(), x.grad.clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
            used[comp] = ("ea_eva_layer_fwd" in calls, "ea_eva_layer_bwd" in calls or "ea_eva_layer_bwd2" in calls)
        pooled_three_nodes = (not module_fn) and dim == 192 and _ops.USE_PROJ_POOL
        assert used[True] == ((False, False) if pooled_three_nodes else (True, True)), used
        assert used[False] == (False, False)
        assert torch.equal(res[True][0], res[False][0])
        assert torch.equal(res[True][1], res[False][1])
        assert res[True][2].keys() == res[False][2].keys()
        for n in res[True][2]:
            a, b = res[True][2][n], res[False][2][n]
            if n == "local_relative_position_bias_table":
                # the single node adds the bias-gradient partials up in its terminal ea_multi_sum launch (another order of
                # additions than ea_colsum_f32), and the general-geometry window backward (window 4 here) accumulates them
                # with fp32 LDS atomics whose order is not fixed from run to run (ea_window_bwd.hip, bias mode 2)
                assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()), n
            else:
                assert torch.equal(a, b), n


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("rows", [100352, 25088, 1000])
def test_wgrad_pair_matches_two_launches(dtype, rows):
    """ea_wgrad_pair (both projections' weight + bias gradients over the same token rows in one launch) against two ea_wgrad
    launches and against an fp64 product: same tiles and MFMA order per slice, different slice boundaries -> fp32 summation
    order differs, nothing else."""
    import torch
    from efficient_attention import _ops
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    g = torch.Generator(device="cuda").manual_seed(3)
    dy1 = torch.randn(rows, 576, device="cuda", generator=g).to(td)
    x1 = torch.randn(rows, 192, device="cuda", generator=g).to(td)
    dy2 = torch.randn(rows, 192, device="cuda", generator=g).to(td)
    x2 = torch.randn(rows, 192, device="cuda", generator=g).to(td)
    _ops.USE_WGRAD_PAIR, keep_switch = True, _ops.USE_WGRAD_PAIR        # (EA_WGRAD_PAIR=0 runs of the suite still test the kernel)
    try:
        assert _ops.wgrad_pair_usable(dy1, x1, dy2, x2)
    finally:
        _ops.USE_WGRAD_PAIR = keep_switch
    (p1, m1), (p2, m2) = _ops.wgrad_pair(dy1, x1, True, dy2, x2, True)
    assert p1.shape[0] == p2.shape[0] and p1.shape[0] < _ops.nv.lib().ea_wgrad_parts(rows, 576, 192) + (1 if rows < 4096 else 0)
    s1, s2 = _ops.multi_sum([p1, p2])
    dw1, db1 = _ops._wgrad_split(s1, m1)
    dw2, db2 = _ops._wgrad_split(s2, m2)
    for (dw, db, dy, x) in ((dw1, db1, dy1, x1), (dw2, db2, dy2, x2)):
        rw, rb = _ops.wgrad(dy, x, True)
        ref_w = dy.double().t() @ x.double()
        ref_b = dy.double().sum(0)
        sc = float(ref_w.abs().max())
        assert float((dw.double() - ref_w).abs().max()) <= 2e-5 * sc + 1e-3
        assert float((rw.double() - ref_w).abs().max()) <= 2e-5 * sc + 1e-3
        assert float((db.double() - ref_b).abs().max()) <= 2e-5 * float(ref_b.abs().max()) + 1e-3
        assert float((dw - rw).abs().max()) <= 2e-5 * sc + 1e-3 and float((db - rb).abs().max()) <= 1e-3 + 2e-5 * float(ref_b.abs().max())
    # one without bias, different widths that still share the tile edge
    (q1, n1), (q2, n2) = _ops.wgrad_pair(dy1, x1, False, dy2, x2, True)
    t1, t2 = _ops.multi_sum([q1, q2])
    assert _ops._wgrad_split(t1, n1)[1] is None
    assert torch.equal(_ops._wgrad_split(t1, n1)[0], dw1) and torch.equal(_ops._wgrad_split(t2, n2)[0], dw2)
    # 512-wide against 192-wide: different tile edges -> not pairable
    assert _ops.wgrad_pair_parts(rows, 1536, 512, 192, 192) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("attn", ["lara", "eva"])
def test_module_backward_with_paired_weight_gradients(attn, monkeypatch):
    """LaraModuleFn / EvaModuleFn with ea_wgrad_pair against the two-launch path: every gradient within fp32 summation noise."""
    import warnings
    import torch
    import efficient_attention as ea
    from efficient_attention import _ops
    args = (dict(dim=192, num_heads=3, num_landmarks=49, proposal_gen="pool-mixed", mis_type="mis-opt", alpha_coeff=2.0)
            if attn == "lara" else dict(dim=192, num_heads=3, num_landmarks=49, window_size=7, attn_2d=True, use_rpe=True,
                                        adaptive_proj="default"))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.manual_seed(16)
        m = ea.AttentionFactory.build_attention(attn, args).cuda()
    m.train()
    x0 = torch.randn(4, 28, 28, 192, device="cuda")
    g = torch.randn(4, 28, 28, 192, device="cuda").bfloat16()
    res, used = {}, {}
    for sw in ("USE_MULTI_SUM", "USE_LARA_MODULE_FN", "USE_EVA_MODULE_FN"):     # what the paired launch builds on
        monkeypatch.setattr(_ops, sw, True)
    for pair in (True, False):
        monkeypatch.setattr(_ops, "USE_WGRAD_PAIR", pair)
        calls = []
        orig = _ops.nv.call

        def spy(name, *a, _calls=calls, _orig=orig):
            _calls.append(name)
            return _orig(name, *a)
        monkeypatch.setattr(_ops.nv, "call", spy)
        for p in m.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        torch.manual_seed(5)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(x)
        y.backward(g)
        monkeypatch.setattr(_ops.nv, "call", orig)
        used[pair] = "ea_wgrad_pair" in calls
        res[pair] = (y.detach().clone(), x.grad.clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
    assert used == {True: True, False: False}
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    for n in res[True][2]:
        a, b = res[True][2][n], res[False][2][n]
        if n in ("qkv.weight", "qkv.bias", "proj.weight", "proj.bias"):
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-6, n
        else:
            assert torch.equal(a, b), n


@pytest.mark.gpu
@pytest.mark.parametrize("attn", ["lara", "eva"])
def test_rounded_weight_from_the_projection_launch(attn, monkeypatch):
    """ea_linear_w32_pool's w_cast output (the rounded qkv weight for the backward's input-gradient GEMM, written by
    workgroup 0 of the projection launch) equals weight.to(dtype) bit for bit, and the module's dx with it equals the dx with
    a cast launch in the backward."""
    import warnings
    import torch
    import efficient_attention as ea
    from efficient_attention import _ops
    args = (dict(dim=192, num_heads=3, num_landmarks=49, proposal_gen="pool-mixed", mis_type="mis-opt", alpha_coeff=2.0)
            if attn == "lara" else dict(dim=192, num_heads=3, num_landmarks=49, window_size=7, attn_2d=True, use_rpe=True,
                                        adaptive_proj="default"))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.manual_seed(17)
        m = ea.AttentionFactory.build_attention(attn, args).cuda()
    m.train()
    x0 = torch.randn(2, 28, 28, 192, device="cuda")
    g = torch.randn(2, 28, 28, 192, device="cuda").to(torch.float16)
    res, seen = {}, []
    for sw in ("USE_PROJ_POOL", "USE_LARA_MODULE_FN", "USE_EVA_MODULE_FN"):     # the path that has the output
        monkeypatch.setattr(_ops, sw, True)
    monkeypatch.setattr(_ops, "USE_W192", False)           # (round 6: the prepared weights replace this output by default)
    orig = _ops.project_qkv_pooled
    for keep in (True, False):
        def wrapped(*a, w_cast=None, _keep=keep, **k):
            out = orig(*a, w_cast=w_cast if _keep else None, **k)
            if _keep and w_cast is not None:
                seen.append(w_cast)
            return out
        monkeypatch.setattr(_ops, "project_qkv_pooled", wrapped)
        if not keep:
            # the module allocated w16 but the kernel did not fill it: make the backward fall back to the cast
            real_mm = _ops._mm_out
            monkeypatch.setattr(_ops, "_mm_out", lambda a, b, dt: real_mm(a, m.qkv.weight.to(a.dtype) if tuple(b.shape) == (576, 192) else b, dt))
        for p in m.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        torch.manual_seed(5)
        with torch.autocast("cuda", dtype=torch.float16):
            y = m(x)
        y.backward(g)
        res[keep] = x.grad.clone()
    assert len(seen) == 1 and torch.equal(seen[0], m.qkv.weight.detach().to(torch.float16))
    assert torch.equal(res[True], res[False])


@pytest.mark.gpu
@pytest.mark.parametrize("xdtype", ["bf16", "fp16", "fp32"])
@pytest.mark.parametrize("rows,C", [(6272, 192), (49, 128), (1000, 512), (37, 1024), (5, 64)])
def test_layernorm_kernels_match_torch(xdtype, rows, C):
    """ea_layernorm_fwd / _bwd (LinearRA's 'dense' generator LayerNorm) against torch.nn.functional.layer_norm evaluated in
    fp64 on the same (rounded) input: output, input gradient, d gamma, d beta."""
    import torch
    import torch.nn.functional as F
    from efficient_attention import _ops
    td = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[xdtype]
    g = torch.Generator(device="cuda").manual_seed(11)
    x = (torch.randn(rows, C, device="cuda", generator=g) * 1.7 + 0.3).to(td).requires_grad_(True)
    w = (1 + 0.2 * torch.randn(C, device="cuda", generator=g)).requires_grad_(True)
    b = (0.1 * torch.randn(C, device="cuda", generator=g)).requires_grad_(True)
    dy = torch.randn(rows, C, device="cuda", generator=g)
    y = _ops.LayerNormFn.apply(x, w, b, 1e-5)
    assert y.dtype == torch.float32
    y.backward(dy)
    xr = x.detach().double().requires_grad_(True)
    wr = w.detach().double().requires_grad_(True)
    br = b.detach().double().requires_grad_(True)
    yr = F.layer_norm(xr, (C,), wr, br, 1e-5)
    yr.backward(dy.double())

    def close(a, ref, tol, what):
        sc = float(ref.abs().max())
        assert float((a.double() - ref).abs().max()) <= tol * sc + 1e-7, (what, float((a.double() - ref).abs().max()) / sc)
    close(y, yr, 2e-6, "y")
    close(x.grad, xr.grad, {"bf16": 8e-3, "fp16": 1e-3, "fp32": 5e-6}[xdtype], "dx")       # dx is rounded to x's type
    close(w.grad, wr.grad, 1e-5, "dgamma")
    close(b.grad, br.grad, 1e-5, "dbeta")


@pytest.mark.gpu
@pytest.mark.parametrize("attn", ["softmax", "local", "local_norpe", "softmax_dropout", "performer"])
def test_core_module_single_node_equals_three_nodes(attn, monkeypatch):
    """CoreModuleFn (softmax / local-window baselines: projections + core as one autograd node, paired weight gradients,
    one terminal reduction) against the three-node path: same kernels for the activations -- y, dx and the bias-table
    gradient bit for bit -- and the projection gradients to fp32 summation order."""
    import warnings
    import torch
    import efficient_attention as ea
    from efficient_attention import _ops
    name = "local" if attn.startswith("local") else ("performer" if attn == "performer" else "softmax")
    args = dict(dim=192, num_heads=3)
    if name == "performer":
        if _ops.PERFORMER_16BIT:
            pytest.skip("EA_PERFORMER_16BIT=1: the single node wraps the fp32 core only")
        args.update(approx_attn_dim=64, proj_method="favorp")
    if name == "local":
        args.update(window_size=7, attn_2d=True, use_rpe=attn == "local")
    if attn == "softmax_dropout":
        args.update(attn_drop=0.1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.manual_seed(18)
        m = ea.AttentionFactory.build_attention(name, args).cuda()
    m.train()
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.02 * torch.randn_like(p))
    if attn == "softmax_dropout":
        gk = torch.Generator().manual_seed(9)
        keep = (torch.rand(4, 3, 196, 196, generator=gk) >= 0.1)
        m._keep_mask_fn = lambda shape: keep
    x0 = torch.randn(4, 14, 14, 192, device="cuda")
    g = torch.randn(4, 14, 14, 192, device="cuda").bfloat16()
    for sw in ("USE_MULTI_SUM", "USE_WGRAD_PAIR", "USE_LARA_MODULE_FN"):
        monkeypatch.setattr(_ops, sw, True)
    res = {}
    for single in (True, False):
        monkeypatch.setattr(_ops, "USE_CORE_MODULE_FN", single)
        for p in m.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        torch.manual_seed(6)                                   # (Performer draws its random features per training call)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(x)
        assert type(y.grad_fn).__name__.startswith("CoreModuleFn") == single
        y.backward(g)
        res[single] = (y.detach().clone(), x.grad.clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    assert res[True][2].keys() == res[False][2].keys()
    for n in res[True][2]:
        a, b = res[True][2][n], res[False][2][n]
        if n.startswith(("qkv.", "proj.")):
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-6, n
        else:
            assert torch.equal(a, b), n


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["rpe_2d_w7", "rpe_2d_w7_e3", "t5_16x32", "t5_128x256"])
def test_table_bias_matches_the_framework_chain(case):
    """_ops.TableBias (ea_table_bias_fwd / _bwd, round 6): the dense window bias out of its table in one launch each way, against
    the framework chain it replaces -- table[index] -> permute -> * log2(e) -> pad -- and that chain's autograd gradient."""
    import math
    import torch
    from efficient_attention import _ops
    from efficient_attention.local_attention import relative_position_index_2d
    from efficient_attention.eva import T5RelativePositionBias
    g = torch.Generator(device="cuda").manual_seed(len(case))
    if case.startswith("rpe"):
        w, e = (7, 0) if case == "rpe_2d_w7" else (7, 3)
        idx = relative_position_index_2d(w, e)
        rows, h, scale = int(idx.max()) + 3, 3, 1.0    # a few table rows nobody reads: their gradient is zero
    else:
        i, j = (16, 32) if case == "t5_16x32" else (128, 256)
        t5 = T5RelativePositionBias(0.125, num_heads=8, causal=False, num_buckets=16 if i == 16 else 64, max_distance=j)
        idx = t5.bucket_table(i, j, torch.device("cpu"))
        rows, h, scale = t5.num_buckets, 8, 0.125
    Wq, Wk = idx.shape
    ld = (Wk + 15) // 16 * 16 + 16
    table = torch.randn(rows, h, device="cuda", generator=g, requires_grad=True)
    tb = _ops.TableBias(idx, rows, Wq, Wk, scale)
    got = tb.dense(table.detach(), ld)
    ref = (table[idx.to("cuda").reshape(-1)].view(Wq, Wk, h).permute(2, 0, 1) * scale) * math.log2(math.e)
    assert got.shape == (h, Wq, ld) and bool((got[..., Wk:] == 0).all())
    assert torch.allclose(got[..., :Wk], ref, rtol=1e-6, atol=1e-7)
    gb = torch.randn(h, Wq, ld, device="cuda", generator=g)
    dt = tb.grad(gb)
    # the kernels' bias gradient is with respect to the natural-unit bias: scale * table[index]
    (ref / math.log2(math.e) * gb[..., :Wk]).sum().backward()
    assert torch.allclose(dt, table.grad, rtol=1e-5, atol=1e-5 * float(table.grad.abs().max()))
    assert torch.equal(dt, tb.grad(gb))                # fixed order
    # a one-column table broadcast over the heads (causal EVA's single-head T5 table): bias rows repeat, gradients add up
    t1 = table.detach()[:, :1].contiguous()
    got1 = tb.dense(t1, ld, heads=h)
    assert torch.equal(got1, tb.dense(t1.expand(rows, h).contiguous(), ld))
    d1 = tb.grad(gb, 1)
    assert d1.shape == (rows, 1) and torch.allclose(d1[:, 0], dt.sum(1), rtol=1e-5, atol=1e-5 * float(dt.abs().max()) * h)
    # ... and with the heads added up before the table kernel sees them (what EvaAttnFn's backward hands over for such a table)
    d1s = tb.grad(gb.sum(0, keepdim=True), 1)
    assert d1s.shape == (rows, 1) and torch.allclose(d1s, d1, rtol=1e-5, atol=1e-5 * float(dt.abs().max()) * h)
    # long position lists are cut into pieces (a far bucket of the 128 x 256 window holds thousands of positions)
    assert (tb._parts > 1) == (case == "t5_128x256" and _ops.TABLE_BIAS_SPLIT)
    if tb._parts > 1:
        import efficient_attention._ops as o
        old = o.TABLE_BIAS_SPLIT
        try:
            o.TABLE_BIAS_SPLIT = False
            tb1 = _ops.TableBias(idx, rows, Wq, Wk, scale)
            whole = tb1.grad(gb)
            assert tb1._parts == 1
        finally:
            o.TABLE_BIAS_SPLIT = old
        assert torch.allclose(dt, whole, rtol=1e-5, atol=1e-5 * float(whole.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("attn", ["eva_rpe", "eva_t5", "local_rpe"])
def test_table_bias_module_path_equals_dense_bias_path(attn, monkeypatch):
    """The single-node module paths with the table handed over (TableBias) against the same modules on the dense-bias chain
    (EA_TABLE_BIAS=0): same kernels downstream, so y and every gradient agree to fp32 rounding of the bias values."""
    import torch
    import efficient_attention as ea
    from efficient_attention import _ops
    kw = dict(dim=192, num_heads=3, qkv_bias=True, attn_drop=0.0, proj_drop=0.0, window_size=7, attn_2d=True,
              overlap_window=False, fp32=False)
    if attn == "eva_rpe":
        kw.update(use_rpe=True, adaptive_proj="default", num_landmarks=49, use_t5_rpe=False)
    elif attn == "eva_t5":
        kw.update(use_rpe=False, adaptive_proj="default", num_landmarks=49, use_t5_rpe=True)
    else:
        kw.update(use_rpe=True)
    torch.manual_seed(3)
    m = ea.AttentionFactory.build_attention("local" if attn == "local_rpe" else "eva", kw).cuda().train()
    x = torch.randn(64, 28, 28, 192, device="cuda")
    gy = torch.randn(64, 28, 28, 192, device="cuda")
    res = []
    for on in (True, False):
        monkeypatch.setattr(_ops, "USE_TABLE_BIAS", on)
        for p in m.parameters():
            p.grad = None
        xi = x.clone().requires_grad_(True)
        torch.manual_seed(11)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(xi)
        y.backward(gy.to(y.dtype))
        res.append([y.float(), xi.grad] + [p.grad.clone() for p in m.parameters()])
    names = ["y", "dx"] + [n for n, _ in m.named_parameters()]
    for n, a, b in zip(names, *res):
        assert torch.allclose(a, b, rtol=2e-3, atol=2e-3 * float(b.abs().max()) + 1e-12), n


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_multi_cast_equals_to_dtype(dtype):
    """ea_multi_cast (round 6): the autocast casts of a layer's parameters in one launch, bit-equal to `.to(dtype)` -- aligned
    bulk, ragged tails, a one-element tensor, values on rounding ties."""
    import torch
    from efficient_attention import _ops
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    g = torch.Generator(device="cuda").manual_seed(7)
    ts = [torch.randn(1536, 512, device="cuda", generator=g) * 0.02, torch.randn(1536, device="cuda", generator=g),
          torch.randn(320, 320, device="cuda", generator=g), torch.randn(333, device="cuda", generator=g) * 100,
          torch.randn(1, device="cuda", generator=g), torch.randn(2051, device="cuda", generator=g)[3:]]
    # exact ties of the 16-bit grid (round to nearest even) and non-finite values
    tie = torch.tensor([1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -8, 1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -11, float("inf"), -0.0, 65519.0, 1e-45],
                       device="cuda")
    ts.append(tie)
    outs = _ops.multi_cast(ts, td)
    for t, o in zip(ts, outs):
        assert o.dtype == td and o.shape == t.shape
        assert torch.equal(o.view(torch.int16), t.to(td).view(torch.int16))


@pytest.mark.gpu
@pytest.mark.parametrize("attn,dim,heads,grid", [("eva", 320, 5, 24), ("softmax", 512, 8, 12), ("local", 320, 5, 24), ("eva", 512, 8, 14)])
def test_wide_module_path_equals_three_node_path(attn, dim, heads, grid, monkeypatch):
    """320 / 512-wide layers as ONE autograd node with library-GEMM projections (round 6: EvaModuleFn / CoreModuleFn with
    module_proj_lib) against the three-node path (LinearFn, core, LinearFn; EA_WIDE_MODULE_FN=0): the same GEMMs and kernels,
    so y and every gradient agree to the rounding of the weight-gradient slice order."""
    import torch
    import efficient_attention as ea
    from efficient_attention import _ops
    w = 8 if grid % 8 == 0 else 7
    kw = dict(dim=dim, num_heads=heads, qkv_bias=True, attn_drop=0.0, proj_drop=0.0, fp32=False)
    if attn != "softmax":
        kw.update(window_size=w, attn_2d=True, overlap_window=False, use_rpe=True)
    if attn == "eva":
        kw.update(adaptive_proj="default", num_landmarks=36 if w == 8 else 49, use_t5_rpe=False)
    torch.manual_seed(5)
    m = ea.AttentionFactory.build_attention(attn, kw).cuda().train()
    x = torch.randn(16, grid, grid, dim, device="cuda")
    gy = torch.randn(16, grid, grid, dim, device="cuda")
    res, nodes = [], []
    for on in (True, False):
        monkeypatch.setattr(_ops, "USE_WIDE_MODULE_FN", on)
        for p in m.parameters():
            p.grad = None
        xi = x.clone().requires_grad_(True)
        torch.manual_seed(11)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(xi)
        nodes.append(type(y.grad_fn).__name__)
        y.backward(gy.to(y.dtype))
        res.append([y.float(), xi.grad] + [p.grad.clone() for p in m.parameters()])
    assert "ModuleFn" in nodes[0] and "ModuleFn" not in nodes[1], nodes
    names = ["y", "dx"] + [n for n, _ in m.named_parameters()]
    for n, a, b in zip(names, *res):
        assert torch.allclose(a, b, rtol=2e-3, atol=2e-3 * float(b.abs().max()) + 1e-12), (n, float((a - b).abs().max()), float(b.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_w192_prepare_and_wsw_projection(dtype):
    """ea_linear_w192_prepare (round 6): the rounded copies of a 192-wide layer's weights in one launch -- bit-equal to
    `.to(dtype)` / its transpose -- and ea_linear_wsw fed by the pre-arranged copy: BIT-identical to ea_linear_w32_pool on the
    fp32 master weight (outputs, rounded copy of x, pooled rows), pooled and plain, 2 x 2 and 4 x 4 cells."""
    import torch
    from efficient_attention import _ops
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    g = torch.Generator(device="cuda").manual_seed(17)
    wq = torch.randn(576, 192, device="cuda", generator=g) * 0.05
    wp = torch.randn(192, 192, device="cuda", generator=g) * 0.05
    bq = torch.randn(576, device="cuda", generator=g) * 0.1
    w16q, wsw, w16p, w16pT = _ops.prepare_w192(wq, wp, td)
    assert torch.equal(w16q, wq.to(td)) and torch.equal(w16p, wp.to(td)) and torch.equal(w16pT, wp.to(td).t().contiguous())
    assert torch.equal(torch.sort(wsw.view(torch.int16).flatten())[0], torch.sort(w16q.view(torch.int16).flatten())[0])
    for (B, H, W, r) in [(8, 28, 28, 4), (16, 14, 14, 2), (3, 8, 8, 2)]:
        x = torch.randn(B * H * W, 192, device="cuda", generator=g)
        L = (H // r) * (W // r)
        pq0, pk0 = torch.empty(B * 3, L, 64, device="cuda"), torch.empty(B * 3, L, 64, device="cuda")
        pq1, pk1 = torch.empty_like(pq0), torch.empty_like(pk0)
        y0, xc0 = _ops.project_qkv_pooled(x, wq, bq, td, True, B, H, W, r, pq0, pk0)
        y1, xc1 = _ops.project_qkv_wsw(x, wsw, w16q, bq, td, True, (B, H, W, r), pq1, pk1)
        assert torch.equal(y0, y1) and torch.equal(xc0, xc1) and torch.equal(pq0, pq1) and torch.equal(pk0, pk1)
    # plain rows: the register-resident kernel from 65 536 rows on, the LDS-resident one below
    for rows in (70000, 5000):
        x = torch.randn(rows, 192, device="cuda", generator=g)
        y0, _ = _ops.ea_linear(x, wq, bq, td, False, elem_dtype=td)
        y1, _ = _ops.project_qkv_wsw(x, wsw, w16q, bq, td, False)
        assert torch.equal(y0, y1)


@pytest.mark.gpu
@pytest.mark.parametrize("attn", ["lara", "eva", "softmax", "local"])
def test_prepared_weight_path_is_bit_identical(attn, monkeypatch):
    """The 192-wide single-node paths with the prepared 16-bit weights (EA_W192_PREPARE, round 6) against the same paths on the
    fp32 master weights: the same rounded operands reach the same kernels, so y and every gradient are BIT-identical."""
    import torch
    import efficient_attention as ea
    from efficient_attention import _ops
    kw = dict(dim=192, num_heads=3, qkv_bias=True, attn_drop=0.0, proj_drop=0.0, fp32=False)
    if attn in ("eva", "local"):
        kw.update(window_size=7, attn_2d=True, overlap_window=False, use_rpe=True)
    if attn == "eva":
        kw.update(adaptive_proj="default", num_landmarks=49, use_t5_rpe=False)
    if attn == "lara":
        kw.update(num_landmarks=49, kernel_size=None, pool_module_type="light", mis_type="mis-opt", proposal_gen="pool-mixed",
                  use_antithetics=False, use_multisample=False, alpha_coeff=2.0)
    torch.manual_seed(5)
    m = ea.AttentionFactory.build_attention(attn, kw).cuda().train()
    for B in (96, 8):                                  # above / below the row count of the register-resident plain projection
        x = torch.randn(B, 28, 28, 192, device="cuda")
        gy = torch.randn(B, 28, 28, 192, device="cuda")
        res = []
        for on in (True, False):
            monkeypatch.setattr(_ops, "USE_W192", on)
            for p in m.parameters():
                p.grad = None
            xi = x.clone().requires_grad_(True)
            torch.manual_seed(11)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m(xi)
            y.backward(gy.to(y.dtype))
            res.append([y, xi.grad] + [p.grad.clone() for p in m.parameters()])
        names = ["y", "dx"] + [n for n, _ in m.named_parameters()]
        for n, a, b in zip(names, *res):
            assert torch.equal(a, b), (attn, B, n, float((a.float() - b.float()).abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols", [(9216, 768), (18816, 384), (65536, 768), (4096, 768), (18, 196608)])
def test_colsum_f32_two_stage_view_matches_sum(rows, cols):
    """_ops.colsum_f32 reads a tall matrix as [rows / k, k * cols] and adds the k partial rows with a second launch
    (_colsum_fold): same sums as one stage, in a fixed order (two calls agree bit for bit)."""
    import torch
    from efficient_attention import _ops
    torch.manual_seed(rows + cols)
    x = torch.randn(rows, cols, device="cuda")
    ref = x.double().sum(0)
    got = _ops.colsum_f32(x)
    assert got.shape == (cols,)
    assert (got.double() - ref).abs().max().item() <= 1e-5 * (rows ** 0.5) * 4
    assert torch.equal(got, _ops.colsum_f32(x))
    one = _ops._colsum_raw(x)
    assert (one.double() - ref).abs().max().item() <= 1e-5 * (rows ** 0.5) * 4
    assert (_ops._colsum_fold(rows, cols) > 1) == (_ops.COLSUM_TWO_STAGE and rows * cols >= (4 << 20) and cols < 8192)


@pytest.mark.gpu
@pytest.mark.parametrize("C,bias", [(512, True), (1024, True), (320, False)])
def test_stacked_linear_equals_linear_on_the_concatenated_weights(C, bias):
    """_ops.StackedLinearFn (causal EVA's q / k / v projections of a wide layer): the 16-bit stacked operand cast straight from
    the three master weights -- same GEMM, same weight-gradient kernel on the same 16-bit operands as LinearFn on
    torch.cat(weights): y, dx and the gradients of all six parameters are identical."""
    import torch
    from efficient_attention import _ops
    if not _ops.USE_STACKED_LINEAR:
        pytest.skip("StackedLinearFn is switched off (EA_STACKED_LINEAR=0)")
    torch.manual_seed(C)
    lins = [torch.nn.Linear(C, C, bias=bias).cuda() for _ in range(3)]
    x = torch.randn(96, 4, C, device="cuda")
    gy = torch.randn(96, 4, 3 * C, device="cuda")
    res = []
    for stacked in (True, False):
        for l in lins:
            l.weight.grad = None
            if bias:
                l.bias.grad = None
        xi = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            if stacked:
                y = _ops.linear_stacked(xi, lins)
                assert y is not None
            else:
                w = torch.cat([l.weight for l in lins], 0)
                b = torch.cat([l.bias for l in lins], 0) if bias else None
                y = _ops.linear_wb(xi, w, b)
        assert y.dtype == torch.bfloat16 and y.shape == (96, 4, 3 * C)
        y.backward(gy.to(y.dtype))
        res.append([y.float(), xi.grad] + [l.weight.grad.clone() for l in lins] + ([l.bias.grad.clone() for l in lins] if bias else []))
    for i, (a, b) in enumerate(zip(*res)):
        assert torch.equal(a, b), i
    # narrow layers stay on this library's own projection kernels
    small = [torch.nn.Linear(128, 128).cuda() for _ in range(3)]
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert _ops.linear_stacked(torch.randn(64, 2, 128, device="cuda"), small) is None


@pytest.mark.gpu
@pytest.mark.parametrize("dim,heads,N,L,variant", [(512, 8, 1024, 64, "plain"), (128, 2, 600, 16, "mask"), (512, 8, 520, 32, "antithetic"),
                                                     (192, 3, 768, 49, "eval")])
def test_lara_1d_module_path_equals_three_node_path(dim, heads, N, L, variant, monkeypatch):
    """LinearRA 'adaptive-1d' as ONE autograd node (round 6: CoreModuleFn around a GraphCore that holds the segment / landmark /
    estimator Functions) against the three-node path (EA_LARA_1D_MODULE_FN=0): the same kernels in the same order, so y is
    identical and every gradient agrees to the rounding of the weight-gradient slice order."""
    import torch
    import efficient_attention as ea
    from efficient_attention import _ops
    if not (_ops.USE_SEGLIN and _ops.USE_CORE_MODULE_FN and _ops.USE_LARA_MODULE_FN and _ops.USE_WIDE_MODULE_FN):
        pytest.skip("the segment kernels or the single-node paths are switched off")
    kw = dict(dim=dim, num_heads=heads, qkv_bias=True, attn_drop=0.0, proj_drop=0.0, num_landmarks=L, proposal_gen="adaptive-1d",
              use_antithetics=variant == "antithetic", fp32=False)
    torch.manual_seed(5)
    m = ea.AttentionFactory.build_attention("lara", kw).cuda()
    m.train(variant != "eval")
    B = 4
    x = torch.randn(B, N, dim, device="cuda")
    gy = torch.randn(B, N, dim, device="cuda")
    mask = None
    if variant == "mask":
        mask = torch.zeros(B, N, dtype=torch.bool, device="cuda")
        mask[1, N - 37:] = True
        mask[3, N - 200:] = True
    res, nodes = [], []
    for on in (True, False):
        monkeypatch.setattr(_ops, "USE_LARA_1D_MODULE_FN", on)
        for p in m.parameters():
            p.grad = None
        xi = x.clone().requires_grad_(True)
        torch.manual_seed(11)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(xi, mask) if mask is not None else m(xi)
        nodes.append(type(y.grad_fn).__name__)
        y.backward(gy.to(y.dtype))
        res.append([y.float(), xi.grad] + [torch.zeros_like(p) if p.grad is None else p.grad.clone() for p in m.parameters()])
    assert "CoreModuleFn" in nodes[0] and "CoreModuleFn" not in nodes[1], nodes
    names = ["y", "dx"] + [n for n, _ in m.named_parameters()]
    assert torch.equal(res[0][0], res[1][0])
    for n, a, b in zip(names, *res):
        assert torch.allclose(a, b, rtol=2e-3, atol=2e-3 * float(b.abs().max()) + 1e-12), (n, float((a - b).abs().max()), float(b.abs().max()))
    # inference: no graph is recorded, same output
    m.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        monkeypatch.setattr(_ops, "USE_LARA_1D_MODULE_FN", True)
        y1 = m(x, mask) if mask is not None else m(x)
        monkeypatch.setattr(_ops, "USE_LARA_1D_MODULE_FN", False)
        y0 = m(x, mask) if mask is not None else m(x)
    assert torch.equal(y1, y0)


@pytest.mark.gpu
@pytest.mark.parametrize("num_samples", [1, -1, 0])
def test_randomized_attention_single_node_path_equals_three_node_path(num_samples, monkeypatch):
    """Randomized attention joins the single-node module path through _ops.GraphCore (round 6: its `_attend`, a chain of this
    library's Functions and framework glue, recorded inside CoreModuleFn's forward): same kernels as the three-node path
    (EA_GRAPH_CORE=0), y identical, gradients to the rounding of the weight-gradient slice order."""
    import torch
    import efficient_attention as ea
    from efficient_attention import _ops
    if not (_ops.USE_CORE_MODULE_FN and _ops.USE_LARA_MODULE_FN):
        pytest.skip("the single-node paths are switched off")
    torch.manual_seed(9)
    m = ea.AttentionFactory.build_attention("ra", dict(dim=192, num_heads=3, qkv_bias=True, attn_drop=0.0, proj_drop=0.0,
                                                       num_samples=num_samples)).cuda().train()
    B, H, W = 8, 14, 14
    x = torch.randn(B, H, W, 192, device="cuda")
    gy = torch.randn(B, H, W, 192, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(3)
    index = torch.randint(0, H * W, (B, 3, H * W), device="cuda", generator=g)
    m._sample_index_fn = lambda shape: index
    res, nodes = [], []
    for on in (True, False):
        monkeypatch.setattr(_ops, "USE_GRAPH_CORE", on)
        for p in m.parameters():
            p.grad = None
        xi = x.clone().requires_grad_(True)
        torch.manual_seed(11)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(xi)
        nodes.append(type(y.grad_fn).__name__)
        y.backward(gy.to(y.dtype))
        res.append([y.float(), xi.grad] + [p.grad.clone() for p in m.parameters()])
    assert "CoreModuleFn" in nodes[0] and "CoreModuleFn" not in nodes[1], nodes
    assert torch.equal(res[0][0], res[1][0])
    names = ["y", "dx"] + [n for n, _ in m.named_parameters()]
    for n, a, b in zip(names, *res):
        assert torch.allclose(a, b, rtol=2e-3, atol=2e-3 * float(b.abs().max()) + 1e-12), (n, float((a - b).abs().max()), float(b.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("attn_2d,overlap", [(True, False), (False, False)])
def test_scatterbrain_single_node_path_equals_three_node_path(attn_2d, overlap, monkeypatch):
    """ScatterBrain's training step through CoreModuleFn + GraphCore (round 6) against its three-node path (EA_GRAPH_CORE=0):
    y identical, gradients (the relative-position table's included) to the rounding of the weight-gradient slice order.
    (The overlapping-window variant is only finite for small keys -- DESIGN 4b -- and is covered, on the single-node path as
    well, by its golden fixtures in test_gpu_modules.py.)"""
    import torch
    import efficient_attention as ea
    from efficient_attention import _ops
    if not (_ops.USE_CORE_MODULE_FN and _ops.USE_LARA_MODULE_FN):
        pytest.skip("the single-node paths are switched off")
    torch.manual_seed(9)
    kw = dict(dim=192, num_heads=3, qkv_bias=True, attn_drop=0.0, proj_drop=0.0, window_size=7 if attn_2d else 8, attn_2d=attn_2d,
              use_rpe=True, overlap_window=overlap, approx_attn_dim=64)
    m = ea.AttentionFactory.build_attention("scatterbrain", kw).cuda().train()
    x = torch.randn(8, 14, 14, 192, device="cuda") if attn_2d else torch.randn(8, 200, 192, device="cuda")
    x = x * (0.3 if overlap else 1.0)                        # (the overlapping variant is only finite for small keys: DESIGN 4b)
    gy = torch.randn_like(x)
    res, nodes = [], []
    for on in (True, False):
        monkeypatch.setattr(_ops, "USE_GRAPH_CORE", on)
        for p in m.parameters():
            p.grad = None
        xi = x.clone().requires_grad_(True)
        torch.manual_seed(11)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(xi)
        y.backward(gy.to(y.dtype))
        assert y.shape == x.shape
        res.append([y.float(), xi.grad] + [torch.zeros_like(p) if p.grad is None else p.grad.clone() for p in m.parameters()])
    names = ["y", "dx"] + [n for n, _ in m.named_parameters()]
    assert torch.equal(res[0][0], res[1][0])
    assert float(res[0][2 + [n for n, _ in m.named_parameters()].index("local_relative_position_bias_table")].abs().max()) > 0
    for n, a, b in zip(names, *res):
        assert torch.isfinite(a).all() and torch.allclose(a, b, rtol=2e-3, atol=2e-3 * float(b.abs().max()) + 1e-12), \
            (n, float((a - b).abs().max()), float(b.abs().max()))
