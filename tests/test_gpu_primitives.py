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
@pytest.mark.parametrize("rows,cols", [(100352, 576), (100352, 192), (1000, 64), (37, 8), (5000, 2048)])
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
