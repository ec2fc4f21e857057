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


@pytest.mark.gpu
@pytest.mark.parametrize("nwin,mask_tail", [(5, 0), (1, 0), (8, 5), (3, 20)])
def test_half_window_overlap_direct_accumulation_matches_scratch_slices(nwin, mask_tail):
    """1-D windows extended by half a window accumulate dk / dv in the I/O dtype across the two colour-class
    launches (class 0 stores, class 1 adds: WinTiling::cdirect) instead of fp32 scratch slices + a finish pass.  That rounds
    twice where the slices rounded once, and relies on class 0 having written every covered token before class 1 runs.
    Compared here against EA_WIN_CDIRECT=0 (a subprocess: the switch is read once per process) at LARGE magnitudes, odd
    window counts, a single window and a padding mask: agreement within two 16-bit ulps of the largest gradient."""
    import os
    import subprocess
    import sys
    import tempfile
    code = r'''
import sys, torch
sys.path.insert(0, sys.argv[1])
import efficient_attention as ea
torch.manual_seed(3)
nwin, mask_tail, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
w, B, h = 16, 2, 2
N = nwin * w
m = ea.AttentionFactory.build_attention("local", dict(dim=128, num_heads=h, window_size=w, overlap_window=True)).cuda()
x = (4.0 * torch.randn(B, N, 128, device="cuda")).requires_grad_(True)
mask = None
if mask_tail:
    mask = torch.zeros(B, N, dtype=torch.bool, device="cuda"); mask[0, N - mask_tail:] = True
with torch.autocast("cuda", dtype=torch.bfloat16):
    y = m(x, mask)
g = 8.0 * torch.randn_like(y)
y.backward(g)
torch.save({"dx": x.grad.cpu(), "dw": m.qkv.weight.grad.cpu()}, out)
'''
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "efficient-attention_amd")
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for flag in ("1", "0"):
            out = os.path.join(td, "r%s.pt" % flag)
            env = dict(os.environ, EA_WIN_CDIRECT=flag)
            r = subprocess.run([sys.executable, "-c", code, pkg, str(nwin), str(mask_tail), out], env=env, capture_output=True,
                               text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-1500:]
            import torch
            res[flag] = torch.load(out)
    for k in ("dx", "dw"):
        a, b = res["1"][k].float(), res["0"][k].float()
        assert torch.isfinite(a).all()
        assert float((a - b).abs().max()) <= 2 * 2.0 ** -8 * float(b.abs().max()), k
