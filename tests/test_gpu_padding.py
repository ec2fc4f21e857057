"""-m gpu: a FULLY padded batch row.

What the reference does (checked by running it: tests/golden/cases.py `*_fullpad`, tools/debug_pad.py):
  * local / EVA fill masked logits with -5e4 and Performer zeroes the padded features: finite everywhere -- pinned on reference
    vectors by the `local_1d_fullpad`, `eva_1d_fullpad`, `performer_1d_fullpad` fixtures (tests/test_gpu_modules.py);
  * softmax, LARA (and ScatterBrain) fill with -inf: the softmax over a row of -inf is NaN, the padded row comes out NaN and,
    because 0 x NaN = NaN in the backward, so do the gradients of that row and of every shared parameter.
The build reproduces both behaviours; this file checks the second kind, which cannot be a fixture comparison: the padded row
is NaN, every other row is finite and BIT-identical to the same row computed without the padded one (rows are independent).
"""
import os
import sys
import warnings

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-attention_amd")]

NAN_ROW = {
    "softmax": dict(dim=128, num_heads=2),
    "lara": dict(dim=128, num_heads=2, num_landmarks=4, proposal_gen="adaptive-1d", mis_type="mis-opt"),
}
FINITE_ROW = {
    "local": dict(dim=128, num_heads=2, window_size=8, attn_2d=False, use_rpe=True),
    "eva": dict(dim=128, num_heads=2, window_size=8, attn_2d=False, use_rpe=True, num_landmarks=3),
    "performer": dict(dim=128, num_heads=2, approx_attn_dim=64, proj_method="favorp"),
    "ra": dict(dim=128, num_heads=2),
}


def _run(attn, args, x, mask):
    import efficient_attention as ea
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.manual_seed(3)
        m = ea.AttentionFactory.build_attention(attn, dict(args)).cuda().eval()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        return m, m(x, mask)


@pytest.mark.gpu
@pytest.mark.parametrize("attn", sorted(NAN_ROW))
def test_minus_inf_variants_return_nan_for_a_fully_padded_row_like_the_reference(attn):
    torch.manual_seed(0)
    x = torch.randn(3, 24, 128, device="cuda")
    mask = torch.zeros(3, 24, dtype=torch.bool, device="cuda")
    mask[1, :] = True
    mask[2, 20:] = True
    _, y = _run(attn, NAN_ROW[attn], x, mask)
    assert torch.isnan(y[1]).all()                       # softmax over a row of -inf (abstract_attention.py:123-127, lara.py:205-211)
    assert torch.isfinite(y[0]).all() and torch.isfinite(y[2]).all()
    keep = torch.tensor([0, 2], device="cuda")
    _, y2 = _run(attn, NAN_ROW[attn], x[keep], mask[keep])
    assert torch.equal(y[keep], y2)                      # the other rows do not see the padded one


@pytest.mark.gpu
@pytest.mark.parametrize("attn", sorted(FINITE_ROW))
def test_finite_fill_variants_stay_finite_with_gradients(attn):
    torch.manual_seed(0)
    x = torch.randn(3, 24, 128, device="cuda", requires_grad=True)
    mask = torch.zeros(3, 24, dtype=torch.bool, device="cuda")
    mask[1, :] = True
    mask[2, 20:] = True
    m, y = _run(attn, FINITE_ROW[attn], x, mask)
    assert torch.isfinite(y).all()
    y.float().square().sum().backward()
    assert torch.isfinite(x.grad).all()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
