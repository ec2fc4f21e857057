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
