"""Host side of the fp32-FAITHFUL cores (round 5; csrc/ea_f32_attn.hip).

Outside torch.autocast the reference computes its attention in fp32 (abstract_attention.py:120-133,
local_attention.py:134-182, eva.py:138-233).  A module of this package called the same way -- fp32 activations, no
autocast -- keeps fp32 end to end: the two Linear layers as fp32 library GEMMs, the core on ea_f32_attn_fwd / _bwd
(gathered windows under a joint softmax, exact fp32 operands on v_mfma_f32_16x16x4_f32).  Covered: the softmax baseline,
local attention and EVA (every adaptive_proj / bias / overlap / 1-D / 2-D variant); the other variants still round to
bf16 with a warning (_ops.to_io_dtype).  EA_F32_CORES=0 switches the path off.

The token tables are the reference's window / chunk partitions (attn_utils.py:155-166,190-210) written as index
arithmetic; -1 marks a slot that leaves the sequence / grid (a zero row that is masked, as pad + as_strided give)."""
import ctypes
import math
import os

import torch

from . import _native as nv

ENABLED = os.environ.get("EA_F32_CORES", "1") == "1"
_TABLES = {}


def usable(x):
    return ENABLED and x.dtype == torch.float32 and x.is_cuda and not torch.is_autocast_enabled()


def _cached(key, build):
    t = _TABLES.get(key)
    if t is None:
        t = build()
        if len(_TABLES) < 512:
            _TABLES[key] = t
    return t


def window_table_1d(n, side, ext, device, left_only=False):
    """[ceil(n / side), side + (ext | 2 ext)] int32: token of slot j of window g = g side - ext + j, -1 outside [0, n)."""
    def build():
        G = -(-n // side)
        width = side + (ext if left_only else 2 * ext)
        tok = torch.arange(G, device=device).view(-1, 1) * side - ext + torch.arange(width, device=device).view(1, -1)
        return torch.where((tok >= 0) & (tok < n), tok, torch.full_like(tok, -1)).to(torch.int32).contiguous()
    return _cached(("1d", n, side, ext, left_only, str(device)), build)


def window_table_2d(H, W, side, ext, device):
    """[(H / side)(W / side), (side + 2 ext)^2] int32 over a row-major H x W grid, -1 outside it."""
    def build():
        t = side + 2 * ext
        y = (torch.arange(H // side, device=device) * side - ext).view(-1, 1, 1, 1) + torch.arange(t, device=device).view(1, 1, -1, 1)
        x = (torch.arange(W // side, device=device) * side - ext).view(1, -1, 1, 1) + torch.arange(t, device=device).view(1, 1, 1, -1)
        ok = (y >= 0) & (y < H) & (x >= 0) & (x < W)
        tok = torch.where(ok, y * W + x, torch.full_like(y * W + x, -1))
        return tok.reshape(-1, t * t).to(torch.int32).contiguous()
    return _cached(("2d", H, W, side, ext, str(device)), build)


def window_tables(attn_2d, seq_shape, side, ext, device):
    """(query table without extension, key table with it) of the windows of one (b, h)."""
    if attn_2d:
        H, W = seq_shape
        return window_table_2d(H, W, side, 0, device), window_table_2d(H, W, side, ext, device)
    n = seq_shape[0]
    return window_table_1d(n, side, 0, device), window_table_1d(n, side, ext, device)


def _t4(t):
    assert t.dim() == 4 and t.stride(3) == 1 and t.dtype == torch.float32, (t.shape, t.stride(), t.dtype)
    return nv.t4(t)


class GatherAttnFn(torch.autograd.Function):
    """out, lse = gathered attention in exact fp32 (ea_f32_attn_fwd / _bwd).
    q [B,H,Nq,D], k, v [B,H,Nk,D]: fp32 views with contiguous channels; ek, ev [B,H,L,D] (extra keys / values shared by all
    groups) or None; bias [Hb, Wq, Wk] (Hb = H or 1) or None; spec: dict(idx_q [G,Wq], idx_k [G,Wk] int32, kmask / qmask
    [B,N] uint8 or None, keep [B,H,Nq,ld] uint8 or None, keep_scale, knorm, neg_inf, zero_masked_v, causal_e, chunk, lm_base, scale).
    Returns out [B,H,Nq,D] (a view of a [B,Nq,H,D] buffer: merging the heads is free) and lse [B,H,Nq]."""

    @staticmethod
    def forward(ctx, q, k, v, ek, ev, bias, spec):
        B, H, Nq, D = q.shape
        Nk = k.shape[2]
        idx_q, idx_k = spec["idx_q"], spec["idx_k"]
        G, Wq = idx_q.shape
        Wk = 0 if idx_k is None else idx_k.shape[1]
        L = 0 if ek is None else ek.shape[2]
        if ek is not None:
            ek, ev = ek.contiguous(), ev.contiguous()
        bias_hs = bias_bs = 0
        if bias is not None:
            # [Hb, Wq, Wk] (Hb = H or 1) shared by the batch, or [B, H, Wq, Wk] (per-token bias of the one-group case)
            bias = bias.float().contiguous()
            assert bias.shape[-2] >= Wq and bias.shape[-1] >= Wk, (bias.shape, Wq, Wk)
            plane = bias.shape[-2] * bias.shape[-1]
            if bias.dim() == 4:
                bias_hs, bias_bs = plane, plane * bias.shape[1]
            elif bias.shape[0] != 1:
                bias_hs = plane
        keep = spec.get("keep")
        g = nv.ea_f32_attn(B, H, Nq, Nk, D, G, Wq, Wk, L, int(spec.get("knorm", 0)) | (2 if spec.get("zero_masked_v") else 0),
                           int(spec.get("neg_inf", 0)), int(spec.get("causal_e", -1)), int(spec.get("chunk", 0)),
                           int(spec.get("lm_base", 0)), 0 if bias is None else bias.shape[-1], bias_hs, bias_bs,
                           0 if keep is None else keep.shape[-1], float(spec.get("keep_scale", 1.0)), float(spec["scale"]))
        # (every token of the sequence is the query slot of exactly one group, so every row of `out` is written; slots of the
        #  table that leave the sequence (-1) write nothing)
        out = torch.empty((B, Nq, H, D), dtype=torch.float32, device=q.device)
        lse = torch.empty((B, H, Nq), dtype=torch.float32, device=q.device)
        stat = torch.empty((B, H, Nq, 2), dtype=torch.float32, device=q.device)     # (row max, row sum): what the backward reads
        ov = out.permute(0, 2, 1, 3)
        tq, tk, tv, to = _t4(q), _t4(k), _t4(v), _t4(ov)
        tek = _t4(ek) if ek is not None else None
        tev = _t4(ev) if ev is not None else None
        nv.call("ea_f32_attn_fwd", ctypes.byref(g), ctypes.byref(tq), ctypes.byref(tk), ctypes.byref(tv),
                ctypes.byref(tek) if tek is not None else None, ctypes.byref(tev) if tev is not None else None,
                nv.ptr(idx_q), nv.ptr(idx_k), nv.ptr(bias), nv.ptr(spec.get("kmask")), nv.ptr(spec.get("qmask")), nv.ptr(keep),
                ctypes.byref(to), nv.ptr(lse), nv.ptr(stat), nv.stream())
        ctx.save_for_backward(q, k, v, ek, ev, bias, ov, stat, idx_q, idx_k, spec.get("kmask"), spec.get("qmask"), keep)
        ctx.g = g
        ctx.set_materialize_grads(False)
        return ov, lse

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout, dlse):
        q, k, v, ek, ev, bias, ov, stat, idx_q, idx_k, kmask, qmask, keep = ctx.saved_tensors
        g = ctx.g
        B, H, Nq, D = q.shape
        Nk = k.shape[2]
        if dout is None:
            dout = torch.zeros_like(ov)
        if dout.stride(3) != 1:
            dout = dout.contiguous()
        dout = dout.float()
        dq = torch.zeros((B, H, Nq, D), dtype=torch.float32, device=q.device)
        dk = torch.zeros((B, H, Nk, D), dtype=torch.float32, device=q.device)
        dv = torch.zeros_like(dk)
        dek = torch.zeros_like(ek) if ek is not None else None
        dev = torch.zeros_like(ev) if ev is not None else None
        dbias = torch.zeros_like(bias) if (bias is not None and ctx.needs_input_grad[5]) else None
        dlse_c = None if dlse is None else dlse.float().contiguous()
        tq, tk, tv, to, tdo, tdq = _t4(q), _t4(k), _t4(v), _t4(ov), _t4(dout), _t4(dq)
        tek = _t4(ek) if ek is not None else None
        tev = _t4(ev) if ev is not None else None
        nv.call("ea_f32_attn_bwd", ctypes.byref(g), ctypes.byref(tq), ctypes.byref(tk), ctypes.byref(tv),
                ctypes.byref(tek) if tek is not None else None, ctypes.byref(tev) if tev is not None else None,
                nv.ptr(idx_q), nv.ptr(idx_k), nv.ptr(bias), nv.ptr(kmask), nv.ptr(qmask), nv.ptr(keep),
                ctypes.byref(to), ctypes.byref(tdo), nv.ptr(stat), nv.ptr(dlse_c), ctypes.byref(tdq), nv.ptr(dk), nv.ptr(dv),
                nv.ptr(dek), nv.ptr(dev), nv.ptr(dbias), nv.stream())
        return dq, dk, dv, dek, dev, dbias, None


class GatherMeanFn(torch.autograd.Function):
    """mean[b,h,c,:] = (1/J) sum_j x[b,h,idx[c][j],:] over the present, unpadded tokens (eva.py:167-181): ea_f32_gather_mean_*."""

    @staticmethod
    def forward(ctx, x, idx, mask_u8):
        B, H, N, D = x.shape
        Cn, J = idx.shape
        mean = torch.empty((B, H, Cn, D), dtype=torch.float32, device=x.device)
        tx = _t4(x)
        nv.call("ea_f32_gather_mean_fwd", B, H, N, D, Cn, J, ctypes.byref(tx), nv.ptr(idx), nv.ptr(mask_u8), nv.ptr(mean), nv.stream())
        ctx.save_for_backward(idx, mask_u8)
        ctx.shape = (B, H, N, D)
        return mean

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dmean):
        idx, mask_u8 = ctx.saved_tensors
        B, H, N, D = ctx.shape
        Cn, J = idx.shape
        dx = torch.zeros((B, H, N, D), dtype=torch.float32, device=dmean.device)
        nv.call("ea_f32_gather_mean_bwd", B, H, N, D, Cn, J, nv.ptr(idx), nv.ptr(mask_u8), nv.ptr(dmean.float().contiguous()),
                nv.ptr(dx), nv.stream())
        return dx, None, None


def _qkv(qkv5):
    """[B,N,3,h,d] fp32 -> q, k, v [B,h,N,d] strided views."""
    assert qkv5.dtype == torch.float32
    p = qkv5.permute(2, 0, 3, 1, 4)
    return p[0], p[1], p[2]


def _mask(mask_u8):
    return None if mask_u8 is None else mask_u8.to(torch.uint8).contiguous()


def softmax_core(qkv5, mask_u8, keep=None, keep_scale=1.0):
    """dropout(softmax(s QK^T, -inf on padded keys)) V (abstract_attention.py:120-133) -> [B,N,h,d]."""
    q, k, v = _qkv(qkv5)
    N, d = q.shape[2], q.shape[3]
    idx = _cached(("all", N, str(q.device)), lambda: torch.arange(N, device=q.device, dtype=torch.int32).view(1, N))
    spec = dict(idx_q=idx, idx_k=idx, kmask=_mask(mask_u8), neg_inf=1, scale=d ** -0.5, keep=keep, keep_scale=keep_scale)
    out, _ = GatherAttnFn.apply(q, k, v, None, None, None, spec)
    return out.permute(0, 2, 1, 3)


def local_core(qkv5, bias, mask_u8, attn_2d, seq_shape, window, ext):
    """Per-window softmax(s QK^T + bias, -5e4 mask) V (local_attention.py:134-182) -> [B,N,h,d]."""
    q, k, v = _qkv(qkv5)
    d = q.shape[3]
    idx_q, idx_k = window_tables(attn_2d, seq_shape, window, ext, q.device)
    spec = dict(idx_q=idx_q, idx_k=idx_k, kmask=_mask(mask_u8), scale=d ** -0.5)
    out, _ = GatherAttnFn.apply(q, k, v, None, None, bias, spec)
    return out.permute(0, 2, 1, 3)


def eva_core(qkv5, bias, noise, mask_u8, attn_2d, seq_shape, window, ext, chunk, mu_fn):
    """EVA's q, k, v -> out core (eva.py:145-227) in fp32: masked chunk means -> mu networks (mu_fn: the module's own
    Linear / LayerNorm layers on the [B,h,L,d] means -> rf_k_bar, mu) -> omega -> beta (a gathered attention of the omega
    rows over their chunks with the key-norm term) -> windows with the control-variate columns under one softmax."""
    q, k, v = _qkv(qkv5)
    B, h, N, d = q.shape
    scale = d ** -0.5
    dev = q.device
    m8 = _mask(mask_u8)
    if attn_2d:
        idx_c = window_table_2d(seq_shape[0], seq_shape[1], chunk, ext, dev)
    else:
        idx_c = window_table_1d(N, chunk, ext, dev)
    Cn = idx_c.shape[0]
    qm = GatherMeanFn.apply(q, idx_c, m8)
    km = GatherMeanFn.apply(k, idx_c, m8)
    rf_k_bar, mu = mu_fn(qm, km)
    omega = mu if noise is None else mu + noise.float()
    rows = _cached(("rows", Cn, str(dev)), lambda: torch.arange(Cn, device=dev, dtype=torch.int32).view(Cn, 1))
    beta, _ = GatherAttnFn.apply(omega.contiguous(), k, v, None, None, None,
                                 dict(idx_q=rows, idx_k=idx_c, kmask=m8, knorm=1, zero_masked_v=1, scale=scale))
    idx_q, idx_k = window_tables(attn_2d, seq_shape, window, ext, dev)
    out, _ = GatherAttnFn.apply(q, k, v, rf_k_bar.contiguous(), beta.contiguous(), bias,
                                dict(idx_q=idx_q, idx_k=idx_k, kmask=m8, scale=scale))
    return out.permute(0, 2, 1, 3)


def _prm(data, proj, scale):
    """prm_projection(normalize=False) (attn_utils.py:324-336,347): s proj.data^T - s |data|^2 / 2 -> [B,h,C,M] (tiny matrices)."""
    return scale * proj @ data.transpose(-1, -2) - 0.5 * scale * (data * data).sum(-1).unsqueeze(-2)


def lara_core(qkv5, mask_u8, q_bar, mu, noise, mis_type, alpha_coeff, mode, scale):
    """LinearRA's estimator (lara.py:187-246) in fp32.  The two contractions over the sequence -- kv_stats = softmax_m(log_proj_k) v
    with its log-sum-exp, and the self-normalised combine out_n = sum_c softmax_c(log alpha + log_proj_q + lse_k - log_prop) kv_c --
    run on the fp32 gathered-attention kernels (queries = the omega rows over all keys with the key-norm term; queries = the
    tokens over the C sample rows with a per-token bias).  The [N x C] weight algebra in between (t = softmax over the sequence,
    alpha, its clamp) and the [C x C] proposal densities are torch element-wise / reduction ops on fp32 tensors: a fidelity path.
    mode: 0 one sample per landmark, 1 antithetic, 2 multi-sample (noise [B,h,2L,d])."""
    q, k, v = _qkv(qkv5)
    B, h, N, d = q.shape
    dev = q.device
    q_bar, mu = q_bar.float(), mu.float()
    dup = False
    if noise is None:
        omega = mu
    elif mode == 2:
        omega, dup = mu.repeat(1, 1, 2, 1) + noise.float(), True
    elif mode == 1:
        omega, dup = torch.cat([mu + noise.float(), mu - noise.float()], -2), True
    else:
        omega = mu + noise.float()
    omega = omega.contiguous()
    C = omega.shape[2]
    m8 = _mask(mask_u8)
    rows_c = _cached(("row", C, str(dev)), lambda: torch.arange(C, device=dev, dtype=torch.int32).view(1, C))
    all_n = _cached(("all", N, str(dev)), lambda: torch.arange(N, device=dev, dtype=torch.int32).view(1, N))
    kv, lse_k = GatherAttnFn.apply(omega, k, v, None, None, None,
                                   dict(idx_q=rows_c, idx_k=all_n, kmask=m8, neg_inf=1, knorm=1, scale=scale))
    if mis_type == "mis-biased":
        lpmu = _prm(mu, omega, scale)
        log_alpha = scale * q @ mu.transpose(-1, -2)                       # [B,h,N,L]
        if dup:
            log_alpha = log_alpha.repeat(1, 1, 1, 2)
        log_prop = torch.logsumexp(lpmu, -1)
    elif mis_type == "mis-opt":
        t = torch.softmax(scale * q @ q_bar.transpose(-1, -2), dim=-2)     # softmax over the SEQUENCE (lara.py:223): [B,h,N,L]
        mu_c = mu
        if dup:
            mu_c, t = mu.repeat(1, 1, 2, 1), t.repeat(1, 1, 1, 2)
        lpmu = _prm(mu_c, omega, scale)                                    # [B,h,C,C]
        log_prop = torch.diagonal(lpmu, dim1=-1, dim2=-2)
        bh = torch.exp(log_prop - torch.logsumexp(lpmu, -1))
        alpha = bh.unsqueeze(-2) + alpha_coeff * (t - t.mean(-1, keepdim=True))
        log_alpha = torch.log(alpha.clamp(min=1e-8))
    elif mis_type == "mis-bh":
        lpmu = _prm(mu, omega, scale)
        log_alpha = None
        log_prop = torch.logsumexp(lpmu, -1)
    else:
        raise NotImplementedError(mis_type)
    cst = (lse_k - log_prop).unsqueeze(-2)                                 # [B,h,1,C]
    bias = cst.expand(B, h, N, C) if log_alpha is None else log_alpha + cst
    out, _ = GatherAttnFn.apply(q, omega, kv.contiguous(), None, None, bias, dict(idx_q=all_n, idx_k=rows_c, scale=scale))
    return out.permute(0, 2, 1, 3)


def causal_eva_core(qkv5, bias, noise, mask_u8, window, ext, chunk, causal, mu_fn, keep=None, keep_scale=1.0):
    """CausalEVAttention's q, k, v -> out core (causal_eva.py:666-783) in fp32: chunks are never extended, the window extension
    lies on the LEFT only, padded queries are masked as well as padded keys, and with `causal` a query sees the local keys up
    to itself and the control variates of the chunks before its own; attention dropout over the Wk + L columns (keep
    [B,h,N,Wk+L], the reference's layout).  qkv5 [B,N,3,h,d] (any strides with contiguous channels)."""
    q, k, v = _qkv(qkv5)
    B, h, N, d = q.shape
    scale = d ** -0.5
    dev = q.device
    m8 = _mask(mask_u8)
    idx_c = window_table_1d(N, chunk, 0, dev)
    Cn = idx_c.shape[0]
    qm = GatherMeanFn.apply(q, idx_c, m8)
    km = GatherMeanFn.apply(k, idx_c, m8)
    rf_k_bar, mu = mu_fn(qm, km)
    omega = mu if noise is None else mu + noise.float()
    rows = _cached(("rows", Cn, str(dev)), lambda: torch.arange(Cn, device=dev, dtype=torch.int32).view(Cn, 1))
    beta, _ = GatherAttnFn.apply(omega.contiguous(), k, v, None, None, None,
                                 dict(idx_q=rows, idx_k=idx_c, kmask=m8, knorm=1, zero_masked_v=1, scale=scale))
    idx_q = window_table_1d(N, window, 0, dev)
    idx_k = window_table_1d(N, window, ext, dev, left_only=True)
    spec = dict(idx_q=idx_q, idx_k=idx_k, kmask=m8, qmask=m8, scale=scale, keep=keep, keep_scale=keep_scale)
    if causal:
        spec.update(causal_e=ext, chunk=chunk)
    out, _ = GatherAttnFn.apply(q, k, v, rf_k_bar.contiguous(), beta.contiguous(), bias, spec)
    return out.permute(0, 2, 1, 3)
