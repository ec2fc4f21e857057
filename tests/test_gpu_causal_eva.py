"""-m gpu: CausalEVAttention (training/evaluation path) beyond the golden fixtures: the wikitext-103
recipe's window/chunk geometry against the CPU oracle (forward, input and parameter gradients),
also at the recipe's head size (d = 128, query-block backward), and the properties the construction promises -- no output
depends on a later token, and a prefix evaluates to the same outputs as the full sequence (the
check the reference runs in its own `__main__`, causal_eva.py:919-949)."""
import argparse
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-attention_amd"), os.path.join(ROOT, "tests")]

RECIPE = dict(window_size=128, chunk_size=8, causal=True, adaptive_proj="qk", use_t5_rpe=True,
              num_chunks=None, overlap_window=False)          # README.md:184


def _build(embed, heads, attn_args, seed=3, dropout=0.0):
    import efficient_attention as ea
    torch.manual_seed(seed)
    m = ea.AttentionFactory.build_attention(
        "causal_eva", dict(embed_dim=embed, num_heads=heads, self_attention=True, dropout=dropout,
                           attn_args=argparse.Namespace(**attn_args))).cuda()
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.02 * torch.randn_like(p))
        if m.rel_pos_bias is not None:
            m.rel_pos_bias.relative_attention_bias.weight.mul_(20.0)
    return m.eval()


def _oracle(m, embed, heads, attn_args, x_bf, mask, g_bf=None, keep=None, p_drop=0.0):
    """oracle.module_forward on batch-first CPU copies; returns y and (if g is given) grads.  With
    `keep` (attention-dropout decisions) the oracle runs in training mode with zero sampling noise."""
    import oracle
    params = {k: v.detach().float().cpu().requires_grad_(g_bf is not None) for k, v in m.state_dict().items()}
    xr = x_bf.detach().float().cpu().requires_grad_(g_bf is not None)
    mr = None if mask is None else mask.cpu()
    y = oracle.module_forward("causal_eva", dict(embed_dim=embed, num_heads=heads, attn_args=attn_args, dropout=p_drop),
                              params, xr, mr, training=keep is not None,
                              noise_fn=lambda shape: torch.zeros(*shape),
                              keep_fn=lambda shape: keep.float().cpu().reshape(shape))
    if g_bf is None:
        return y.detach(), None, None
    (y * g_bf.float().cpu()).sum().backward()
    return y.detach(), xr.grad, {k: v.grad for k, v in params.items()}


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("variant", ["recipe_d64", "recipe_d128", "overlap_mask", "recipe_d128_dropout",
                                     "overlap_dropout", "overlap_w128_d64", "overlap_w64_d128", "many_chunks"])
def test_recipe_geometry_matches_oracle(variant, dtype):
    from gpu_checks import MODULE_TOL, FP16_TOL
    from util import scaled_err
    from gpu_checks import tol_for
    base = tol_for("causal_eva", dtype, "test_gpu_causal_eva")
    dtype = torch.bfloat16 if dtype == "bf16" else torch.float16
    p_drop = 0.1 if variant.endswith("dropout") else 0.0      # transformer_lm_wiki103's attention dropout
    if variant == "recipe_d64":
        embed, heads, T, B, aa, pads = 512, 8, 512, 4, dict(RECIPE), None
    elif variant == "recipe_d128_dropout":
        embed, heads, T, B, aa, pads = 1024, 8, 512, 2, dict(RECIPE), None
    elif variant == "overlap_w128_d64":
        # overlapping 128-token windows (256 keys): 4 query blocks x 2 colour classes, launched in turn
        embed, heads, T, B, aa, pads = 512, 8, 512, 2, dict(RECIPE, overlap_window=True), [0, 40]
    elif variant == "overlap_w64_d128":
        embed, heads, T, B, aa, pads = 1024, 8, 512, 2, dict(RECIPE, window_size=64, overlap_window=True), None
    elif variant == "overlap_dropout":
        embed, heads, T, B, pads = 128, 2, 100, 2, [0, 11]
        aa = dict(RECIPE, window_size=24, chunk_size=8, overlap_window=True)   # Wk = 48, L = 15: padded mask columns
    elif variant == "many_chunks":
        # 80 chunks > the 64 landmark rows of the 16-bit window kernels: the generic fp32 kernels on the 16-bit activations
        embed, heads, T, B, pads = 256, 4, 310, 2, [0, 23]
        aa = dict(RECIPE, window_size=32, chunk_size=4, overlap_window=True)
    elif variant == "recipe_d128":
        # transformer_lm_wiki103: embed 1024, 8 heads.  128 queries x 128 keys at d = 128 exceed one
        # LDS image in backward: the window runs as 4 query blocks (ea_window_bwd_query_blocks)
        embed, heads, T, B, aa, pads = 1024, 8, 512, 2, dict(RECIPE), [0, 70]
    else:
        embed, heads, T, B, pads = 256, 4, 500, 3, [0, 37, 0]
        aa = dict(RECIPE, window_size=32, chunk_size=16, overlap_window=True, adaptive_proj="no-ln")
    m = _build(embed, heads, aa, dropout=p_drop)
    gen = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(B, T, embed, device="cuda", generator=gen).requires_grad_(True)
    g = torch.randn(B, T, embed, device="cuda", generator=gen)
    keep = None
    if p_drop:
        # training mode with the sampling noise silenced and the dropout decisions shared with the oracle
        w, e = aa["window_size"], (aa["window_size"] if aa["overlap_window"] else 0)
        n = -(-T // w) * w
        keep = (torch.rand(B, heads, n, w + e + n // aa["chunk_size"], device="cuda", generator=gen) >= p_drop)
        m.train()
        m._keep_mask_fn = lambda shape: keep.reshape(shape)
    mask = None
    if pads is not None:
        mask = torch.zeros(B, T, dtype=torch.bool, device="cuda")
        for b, k in enumerate(pads):
            if k:
                mask[b, T - k:] = True
    real_randn_like = torch.randn_like
    if p_drop:
        torch.randn_like = lambda t, **kw: torch.zeros_like(t)
    try:
        with torch.autocast("cuda", dtype=dtype):
            xt = x.transpose(0, 1)
            y = m(xt, xt, xt, key_padding_mask=mask)[0].transpose(0, 1)
    finally:
        torch.randn_like = real_randn_like
    (y.float() * g).sum().backward()
    yr, dxr, pgr = _oracle(m, embed, heads, aa, x, mask, g, keep, p_drop)
    errs = {"y": scaled_err(y.detach().float().cpu().numpy(), yr.numpy()),
            "dx": scaled_err(x.grad.cpu().numpy(), dxr.numpy())}
    for k, p in m.named_parameters():
        errs["d" + k] = scaled_err(p.grad.float().cpu().numpy(), pgr[k].numpy())
    # a constant added to every key shifts all local logits of a query alike, so d k_proj.bias is the
    # small remainder of a cancelling sum over all tokens (only the mask and chunk terms break the
    # symmetry): its rounding error is judged on a 4x wider band (it shrinks 8x from bf16 to fp16
    # like every other figure, which a logic error would not)
    def tol(k):
        return tuple((4.0 if k == "dk_proj.bias" else 1.0) * t for t in base)
    bad = {k: v for k, v in errs.items() if not (v[0] <= tol(k)[0] and v[1] <= tol(k)[1])}
    assert not bad, (variant, bad)


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", [False, True])
def test_no_dependence_on_later_tokens(overlap):
    """Changing tokens >= t0 leaves every output before t0 bit-identical, and the gradient of a loss
    on outputs before t0 with respect to tokens >= t0 is exactly zero."""
    aa = dict(RECIPE, window_size=32, chunk_size=8, overlap_window=overlap)
    m = _build(128, 2, aa)
    gen = torch.Generator(device="cuda").manual_seed(2)
    x = torch.randn(2, 160, 128, device="cuda", generator=gen)
    x2 = x.clone()
    t0 = 77
    x2[:, t0:] = torch.randn(2, 160 - t0, 128, device="cuda", generator=gen)

    def run(inp):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            xt = inp.transpose(0, 1)
            return m(xt, xt, xt)[0].transpose(0, 1)
    with torch.no_grad():
        ya, yb = run(x), run(x2)
    assert torch.equal(ya[:, :t0], yb[:, :t0])
    assert not torch.equal(ya[:, t0:], yb[:, t0:])
    xg = x.clone().requires_grad_(True)
    run(xg)[:, :t0].float().square().sum().backward()
    assert xg.grad[:, t0:].abs().max().item() == 0.0
    assert xg.grad[:, :t0].abs().max().item() > 0.0


@pytest.mark.gpu
def test_prefix_consistency():
    """Outputs of a prefix equal those of the full sequence at the same positions (chunk_size
    given, so the chunking does not depend on the length): causal_eva.py:919-949."""
    aa = dict(RECIPE, window_size=64, chunk_size=16)
    m = _build(128, 2, aa)
    x = torch.randn(4, 512, 128, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4))

    def run(inp):
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            xt = inp.transpose(0, 1)
            return m(xt, xt, xt)[0].transpose(0, 1).float()
    full = run(x)
    scale = full.abs().max().item()
    for n in (26, 64, 100, 257):
        part = run(x[:, :n])
        assert part.shape == (4, n, 128)
        # different GEMM shapes round differently in bf16; the attention itself sees the same rows
        assert (part - full[:, :n]).abs().max().item() <= 2e-2 * scale, n


@pytest.mark.gpu
def test_unsupported_backward_geometry_is_loud():
    """256 keys at d = 128 do not fit the backward kernel's LDS image even with 32-query blocks: it
    must raise, not fall back."""
    m = _build(1024, 8, dict(RECIPE, window_size=256, chunk_size=16))
    x = torch.randn(1, 512, 1024, device="cuda", requires_grad=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        xt = x.transpose(0, 1)
        y = m(xt, xt, xt)[0]
    with pytest.raises(RuntimeError, match="unsupported geometry"):
        y.float().sum().backward()


@pytest.mark.gpu
def test_separate_key_value_inputs():
    """self_attention=False with key = value = query goes through three projections and a stacked
    buffer; it must agree with the fused self-attention path on the same weights."""
    import efficient_attention as ea
    aa = dict(RECIPE, window_size=32, chunk_size=8)
    m1 = _build(128, 2, aa)
    torch.manual_seed(3)
    m2 = ea.AttentionFactory.build_attention(
        "causal_eva", dict(embed_dim=128, num_heads=2, self_attention=False,
                           attn_args=argparse.Namespace(**aa))).cuda().eval()
    m2.load_state_dict(m1.state_dict())
    x = torch.randn(100, 3, 128, device="cuda", generator=torch.Generator(device="cuda").manual_seed(8))
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y1 = m1(xa, xa, xa)[0]
        y2 = m2(xb, xb, xb)[0]
    assert torch.equal(y1, y2)
    y1.float().square().sum().backward()
    y2.float().square().sum().backward()
    scale = xa.grad.abs().max().item()
    assert (xa.grad - xb.grad).abs().max().item() <= 2e-2 * scale      # three bf16 dgrad GEMMs summed vs one
    for (k, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert (p1.grad - p2.grad).abs().max().item() <= 2e-2 * max(p1.grad.abs().max().item(), 1e-6), k


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["recipe_d64", "recipe_d128", "overlap_d64", "no_rpe_noln", "many_chunks"])
def test_incremental_decoding_equals_full_forward(variant):
    """Token-by-token decoding with fairseq's incremental state (reference causal_eva.py:537-665, dead code there: `N`
    unbound) reproduces the pinned full-sequence causal path row by row -- the definition this build gives it
    (causal_eva.py::_decode).  Checked on the recipe geometry (w = 128, chunks of 8, T5 bias) at d = 64 and d = 128, with
    left-extended windows, without the bias / LayerNorm, and beyond 64 chunks; chunks of several steps at once; beam reordering."""
    aa = dict(RECIPE)
    embed, heads, T, B = 512, 8, 300, 2
    if variant == "recipe_d128":
        embed, heads, T = 1024, 8, 200
    elif variant == "overlap_d64":
        aa.update(overlap_window=True, window_size=32)
        T = 150
    elif variant == "no_rpe_noln":
        aa.update(use_t5_rpe=False, adaptive_proj="no-ln", window_size=64, chunk_size=16)
        T = 200
    elif variant == "many_chunks":
        # 80 chunks: past the 64 landmark rows of the 16-bit window kernels both the full path and the decoding steps run
        # the generic fp32 kernels on the 16-bit rows (round 5; no length limit left)
        aa.update(overlap_window=True, window_size=32, chunk_size=4)
        embed, heads, T = 256, 4, 300
    m = _build(embed, heads, aa)
    torch.manual_seed(11)
    x = torch.randn(T, B, embed, device="cuda")
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        full, _ = m(x, x, x)
        state = {}
        m.init_incremental_state()
        rows = []
        t = 0
        for step in (1, 1, 1, 5, 1, 2, 64, 1, 1, 7):             # single tokens and several at once
            while t < T and step:
                n = min(step, T - t)
                y, _ = m(x[t:t + n], x[t:t + n], x[t:t + n], incremental_state=state)
                rows.append(y)
                t += n
                break
        while t < T:
            y, _ = m(x[t:t + 1], x[t:t + 1], x[t:t + 1], incremental_state=state)
            rows.append(y)
            t += 1
    inc = torch.cat(rows, 0)
    assert inc.shape == full.shape
    err = (inc.float() - full.float()).abs().max().item()
    ref = full.float().abs().max().item()
    assert err <= 2e-2 * ref, (variant, err, ref)               # same kernels on the same rows: bf16 rounding of `out` only


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["recipe_d64", "overlap_d64", "many_chunks"])
def test_incremental_decoding_with_padded_positions(variant):
    """`key_padding_mask` during decoding (left-padded prompts of unequal length in one batch): every non-padded row equals
    the row of the full-sequence forward given the same mask.  Batch element 1 starts with 19 padded positions -- two whole
    chunks of 8 (their landmarks are built from no row at all) and part of a third --, element 2 with 3; the mask arrives in
    both shapes fairseq uses (flags of the new positions / of every position so far)."""
    aa = dict(RECIPE)
    embed, heads, T, B = 512, 8, 200, 3
    if variant == "overlap_d64":
        aa.update(overlap_window=True, window_size=32)
        T = 150
    elif variant == "many_chunks":
        aa.update(overlap_window=True, window_size=32, chunk_size=4)
        embed, heads, T = 256, 4, 300
    m = _build(embed, heads, aa)
    torch.manual_seed(13)
    x = torch.randn(T, B, embed, device="cuda")
    pad = torch.zeros(B, T, dtype=torch.bool, device="cuda")
    pad[1, :19] = True
    pad[2, :3] = True
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        full, _ = m(x, x, x, key_padding_mask=pad)
        state = {}
        m.init_incremental_state()
        rows = []
        t = 0
        for i, step in enumerate([1, 1, 6, 1, 20, 1, 2, 64] + [1] * T):
            if t >= T:
                break
            n = min(step, T - t)
            kpm = pad[:, t:t + n] if i % 2 == 0 else pad[:, :t + n]
            y, _ = m(x[t:t + n], x[t:t + n], x[t:t + n], key_padding_mask=kpm, incremental_state=state)
            rows.append(y)
            t += n
    inc = torch.cat(rows, 0)
    assert inc.shape == full.shape
    live = (~pad).t().unsqueeze(-1).float()                      # [T, B, 1]
    err = ((inc.float() - full.float()).abs() * live).max().item()
    ref = (full.float().abs() * live).max().item()
    assert err <= 2e-2 * ref, (variant, err, ref)
    # an unpadded batch given an all-False mask takes the same steps as one given none
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        s1, s2 = {}, {}
        m.init_incremental_state()
        none = torch.zeros(B, 1, dtype=torch.bool, device="cuda")
        for t in range(40):
            a, _ = m(x[t:t + 1], x[t:t + 1], x[t:t + 1], incremental_state=s1)
            b, _ = m(x[t:t + 1], x[t:t + 1], x[t:t + 1], key_padding_mask=none, incremental_state=s2)
            assert (a.float() - b.float()).abs().max().item() <= 2e-2 * a.float().abs().max().item()


@pytest.mark.gpu
def test_incremental_state_reorders_with_the_beam():
    aa = dict(RECIPE, window_size=32)
    m = _build(256, 4, aa)
    torch.manual_seed(5)
    x = torch.randn(40, 3, 256, device="cuda")
    order = torch.tensor([2, 0, 0], device="cuda")
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        st = {}
        m.init_incremental_state()
        for t in range(20):
            m(x[t:t + 1], x[t:t + 1], x[t:t + 1], incremental_state=st)
        m.reorder_incremental_state(st, order)
        xr = x[:, order]
        ys = [m(xr[t:t + 1], xr[t:t + 1], xr[t:t + 1], incremental_state=st)[0] for t in range(20, 40)]
        full, _ = m(xr, xr, xr)
    inc = torch.cat(ys, 0)
    assert (inc.float() - full[20:].float()).abs().max().item() <= 2e-2 * full.float().abs().max().item()


@pytest.mark.gpu
def test_incremental_decoding_refuses_what_it_cannot_pin():
    aa = dict(RECIPE, causal=False)
    m = _build(256, 4, aa)
    x = torch.randn(1, 2, 256, device="cuda")
    with pytest.raises(NotImplementedError):
        m(x, x, x, incremental_state={})


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["recipe_d64", "recipe_d128_dropout", "overlap_mask_noln", "narrow_d64"])
def test_single_node_path_equals_three_node_path(variant, monkeypatch):
    """The training step of the self-attention module as ONE autograd node (round 6: _ops.CoreModuleFn around a GraphCore holding
    the causal EVA core) against the three-node path (EA_CAUSAL_MODULE_FN=0): the same kernels in the same order -- y identical,
    gradients to the rounding of the weight-gradient slice order; wide layers (library GEMMs inside the node) and a narrow one
    (this library's own projection kernels), T5 table, attention dropout with fixed keep decisions, a pad mask."""
    from efficient_attention import _ops
    if not (_ops.USE_CORE_MODULE_FN and _ops.USE_LARA_MODULE_FN and _ops.USE_WIDE_MODULE_FN):
        pytest.skip("the single-node paths are switched off")
    aa = dict(RECIPE)
    embed, heads, T, B, dropout = 512, 8, 256, 3, 0.0
    if variant == "recipe_d128_dropout":
        embed, T, dropout = 1024, 384, 0.1
    elif variant == "overlap_mask_noln":
        aa.update(overlap_window=True, window_size=32, adaptive_proj="no-ln", use_t5_rpe=False)
        T = 150
    elif variant == "narrow_d64":
        aa.update(window_size=32)
        embed, heads, T = 128, 2, 128
    m = _build(embed, heads, aa, dropout=dropout).train()
    torch.manual_seed(21)
    x = torch.randn(T, B, embed, device="cuda")
    gy = torch.randn(T, B, embed, device="cuda")
    mask = None
    if variant == "overlap_mask_noln":
        mask = torch.zeros(B, T, dtype=torch.bool, device="cuda")
        mask[1, T - 21:] = True
    keeps = {}

    def keep_fn(shape):                                  # the same dropout decisions on both paths
        if shape not in keeps:
            g = torch.Generator(device="cuda").manual_seed(7)
            keeps[shape] = (torch.rand(shape, device="cuda", generator=g) >= dropout).to(torch.uint8)
        return keeps[shape]
    m._keep_mask_fn = keep_fn
    res, nodes = [], []
    for on in (True, False):
        monkeypatch.setattr(_ops, "USE_CAUSAL_MODULE_FN", on)
        for p in m.parameters():
            p.grad = None
        xi = x.clone().requires_grad_(True)
        torch.manual_seed(11)                            # the landmark noise
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y, _ = m(xi, xi, xi, key_padding_mask=mask)
        nodes.append(type(y.grad_fn).__name__)
        y.backward(gy.to(y.dtype))
        res.append([y.float(), xi.grad] + [torch.zeros_like(p) if p.grad is None else p.grad.clone() for p in m.parameters()])
    names = ["y", "dx"] + [n for n, _ in m.named_parameters()]
    assert torch.equal(res[0][0], res[1][0]), float((res[0][0] - res[1][0]).abs().max())
    for n, a, b in zip(names, *res):
        assert torch.allclose(a, b, rtol=2e-3, atol=2e-3 * float(b.abs().max()) + 1e-12), (n, float((a - b).abs().max()), float(b.abs().max()))
    # the node really is the single one (y is a slice / contiguous copy of its output)
    from efficient_attention.causal_eva import CausalEVAttention
    calls = []
    orig = _ops.CoreModuleFn.apply
    monkeypatch.setattr(_ops, "USE_CAUSAL_MODULE_FN", True)
    monkeypatch.setattr(_ops.CoreModuleFn, "apply", staticmethod(lambda *a, **k: (calls.append(1), orig(*a, **k))[1]))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        m(x, x, x, key_padding_mask=mask)
    assert calls == [1]
