"""-m gpu: a sweep of window-kernel geometries -- window / extension / landmark / head sizes that
select different backward plans (single launch, colour-class slices, ordered or merged query blocks,
bias table in LDS or global memory) -- module forward + input gradient against the CPU oracle."""
import argparse
import os
import sys
import warnings

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-attention_amd"), os.path.join(ROOT, "tests")]

# (name, attn, x_shape (batch-first), ctor args, trailing pads per batch row or None)
SWEEP = [
    ("local_1d_w8", "local", (2, 100, 64), dict(dim=64, num_heads=2, window_size=8, use_rpe=True), [0, 13]),
    ("local_1d_w8_overlap_d32", "local", (2, 100, 64), dict(dim=64, num_heads=2, window_size=8, use_rpe=True,
                                                            overlap_window=True), [0, 13]),
    ("local_1d_w48_overlap", "local", (2, 200, 128), dict(dim=128, num_heads=2, window_size=48, use_rpe=True,
                                                          overlap_window=True), None),
    ("local_2d_w4_overlap", "local", (2, 16, 16, 128), dict(dim=128, num_heads=2, window_size=4, attn_2d=True,
                                                            use_rpe=True, overlap_window=True), None),
    ("local_2d_w7_d128", "local", (2, 14, 14, 256), dict(dim=256, num_heads=2, window_size=7, attn_2d=True, use_rpe=True), None),
    ("eva_1d_w32_L16", "eva", (2, 512, 128), dict(dim=128, num_heads=2, window_size=32, use_t5_rpe=True, num_landmarks=16,
                                                  adaptive_proj="default"), [0, 40]),
    ("eva_1d_w32_overlap_noln", "eva", (2, 512, 128), dict(dim=128, num_heads=2, window_size=32, use_t5_rpe=True,
                                                           overlap_window=True, num_landmarks=8, adaptive_proj="no-ln"), [0, 40]),
    ("eva_1d_w64_d32_none", "eva", (2, 256, 64), dict(dim=64, num_heads=2, window_size=64, use_rpe=True, num_landmarks=4,
                                                      adaptive_proj="none"), None),
    ("eva_2d_w4_overlap_L16", "eva", (2, 16, 16, 128), dict(dim=128, num_heads=2, window_size=4, attn_2d=True, use_rpe=True,
                                                            overlap_window=True, num_landmarks=16), None),
    ("eva_2d_w8_d128_L4", "eva", (2, 16, 16, 256), dict(dim=256, num_heads=2, window_size=8, attn_2d=True, use_t5_rpe=True,
                                                        num_landmarks=4, adaptive_proj="no-ln"), None),
    ("causal_w16_c8", "causal_eva", (2, 128, 128), dict(window_size=16, chunk_size=8, causal=True, adaptive_proj="qk",
                                                        use_t5_rpe=True, num_chunks=None, overlap_window=False), [0, 9]),
    ("causal_w16_c2_overlap", "causal_eva", (2, 128, 128), dict(window_size=16, chunk_size=2, causal=True, adaptive_proj="no-ln",
                                                                use_t5_rpe=False, num_chunks=None, overlap_window=True), None),
    ("causal_w256_c8_d64", "causal_eva", (2, 512, 128), dict(window_size=256, chunk_size=8, causal=True, adaptive_proj="qk",
                                                             use_t5_rpe=True, num_chunks=None, overlap_window=False), [0, 30]),
    ("causal_off_w64_chunks16", "causal_eva", (2, 256, 128), dict(window_size=64, chunk_size=None, causal=False, adaptive_proj="qk",
                                                                  use_t5_rpe=True, num_chunks=16, overlap_window=True), [0, 30]),
    ("causal_w128_c4_d32", "causal_eva", (2, 256, 64), dict(window_size=128, chunk_size=4, causal=True, adaptive_proj="qk",
                                                            use_t5_rpe=True, num_chunks=None, overlap_window=False), None),
]


def _build(attn, args):
    import efficient_attention as ea
    torch.manual_seed(9)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if attn == "causal_eva":
            embed = args.pop("embed")
            m = ea.AttentionFactory.build_attention(attn, dict(embed_dim=embed, num_heads=2, self_attention=True,
                                                               attn_args=argparse.Namespace(**args)))
        else:
            m = ea.AttentionFactory.build_attention(attn, dict(args))
    m = m.cuda().eval()
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.05 * torch.randn_like(p))
    return m


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("case", SWEEP, ids=[c[0] for c in SWEEP])
def test_geometry_matches_oracle(case, dtype):
    import oracle
    from gpu_checks import MODULE_TOL, FP16_TOL
    from util import scaled_err
    name, attn, shape, args, pads = case
    args = dict(args)
    from gpu_checks import tol_for
    tol = tol_for("causal_eva" if attn == "causal_eva" else attn, dtype, "test_gpu_window_sweep")
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    if attn == "causal_eva":
        args["embed"] = shape[-1]
    m = _build(attn, dict(args))
    gen = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(*shape, device="cuda", generator=gen).requires_grad_(True)
    g = torch.randn(*shape, device="cuda", generator=gen)
    mask = None
    if pads is not None:
        n = shape[1]
        mask = torch.zeros(shape[0], n, dtype=torch.bool, device="cuda")
        for b, k in enumerate(pads):
            if k:
                mask[b, n - k:] = True
    with torch.autocast("cuda", dtype=td):
        if attn == "causal_eva":
            xt = x.transpose(0, 1)
            y = m(xt, xt, xt, key_padding_mask=mask)[0].transpose(0, 1)
        else:
            y = m(x, mask) if mask is not None else m(x)
    (y.float() * g).sum().backward()
    params = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    xr = x.detach().cpu().requires_grad_(True)
    if attn == "causal_eva":
        args.pop("embed")
        oargs = dict(embed_dim=shape[-1], num_heads=2, attn_args=args)
    else:
        oargs = args
    yr = oracle.module_forward(attn, oargs, params, xr, None if mask is None else mask.cpu(), training=False)
    (yr * g.cpu()).sum().backward()
    for what, got, want in (("y", y.detach().float().cpu(), yr.detach()), ("dx", x.grad.cpu(), xr.grad)):
        e = scaled_err(got.numpy(), want.numpy())
        assert e[0] <= tol[0] and e[1] <= tol[1], (name, what, e)
