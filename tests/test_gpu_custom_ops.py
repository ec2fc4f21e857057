"""-m gpu: the dispatcher-registered custom ops (torch.ops.ea.*, efficient_attention/_dispatch.py).

  * every forward op is callable through torch.ops with raw tensors and matches the nn.Module path bit for bit;
  * schema and fake-tensor (meta) implementations agree with the real ones (torch.library.opcheck);
  * a module traced by torch.compile (aot_eager backend: graph capture through autograd.Function + the
    opaque ops, no code generation) reproduces eager output and gradients."""
import os
import sys
import warnings

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-attention_amd")]


def _qkv(B=2, N=196, h=2, d=64, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (0.5 * torch.randn(B, N, 3, h, d, device="cuda", generator=g)).to(torch.bfloat16)


@pytest.mark.gpu
def test_ops_are_registered_with_schemas():
    import efficient_attention  # noqa: F401
    for name in ("softmax", "local", "eva", "lara", "performer"):
        for d in ("fwd", "bwd"):
            op = getattr(torch.ops.ea, "%s_%s" % (name, d))
            assert "ea::%s_%s" % (name, d) in str(op.default._schema)


@pytest.mark.gpu
def test_opcheck_forward_ops():
    import efficient_attention  # noqa: F401
    qkv = _qkv()
    B, N, _, h, d = qkv.shape
    checks = ("test_schema", "test_faketensor")
    torch.library.opcheck(torch.ops.ea.softmax_fwd.default, (qkv, None, None, 1.0), test_utils=checks)
    bias = torch.randn(h, 49, 49, device="cuda")
    torch.library.opcheck(torch.ops.ea.local_fwd.default, (qkv, bias, None, [1, 14, 14, 7, 0]), test_utils=checks)
    W = torch.randn(h, 64, d, device="cuda")
    torch.library.opcheck(torch.ops.ea.performer_fwd.default, (qkv, None, W), test_utils=checks)
    params = []
    for _ in range(2):
        params += [0.1 * torch.randn(d, d, device="cuda"), torch.zeros(d, device="cuda"), torch.ones(d, device="cuda"),
                   torch.zeros(d, device="cuda")]
    noise = torch.randn(B, h, 49, d, device="cuda")
    torch.library.opcheck(torch.ops.ea.lara_fwd.default,
                          (qkv, None, noise, [14, 14, 2, 1, 1, 0, 0, 1], [2.0, d ** -0.5], params), test_utils=checks)
    torch.library.opcheck(torch.ops.ea.eva_fwd.default,
                          (qkv, bias, noise[:, :, :4], None, None, [1, 14, 14, 7, 0, 7, 4, 0, 1], [0.5, 1.0], "default", params),
                          test_utils=checks)


@pytest.mark.gpu
@pytest.mark.parametrize("attn,args", [
    ("lara", dict(dim=128, num_heads=2, num_landmarks=49, proposal_gen="pool-mixed", mis_type="mis-opt", alpha_coeff=2.0)),
    ("eva", dict(dim=128, num_heads=2, window_size=7, attn_2d=True, use_rpe=True, num_landmarks=4, adaptive_proj="default")),
    ("softmax", dict(dim=128, num_heads=2)),
])
def test_compiled_module_matches_eager(attn, args):
    import efficient_attention as ea
    torch.manual_seed(3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = ea.AttentionFactory.build_attention(attn, dict(args)).cuda().eval()
    x = torch.randn(2, 14, 14, 128, device="cuda")
    gy = torch.randn(2, 14, 14, 128, device="cuda")

    def run(fn):
        xx = x.clone().requires_grad_(True)
        for p in m.parameters():
            p.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = fn(xx)
        (y.float() * gy).sum().backward()
        return y.detach().float(), xx.grad.clone(), m.qkv.weight.grad.clone()
    ref = run(m)
    got = run(torch.compile(m, backend="aot_eager"))
    for a, b in zip(ref, got):
        assert torch.allclose(a, b, rtol=1e-3, atol=1e-4), (attn, (a - b).abs().max())
