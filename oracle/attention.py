"""CPU restatement of the attention cores and module forwards (test infrastructure; see
oracle/__init__.py).  Written from the math of SURVEY.md 3 with gather-based windowing;
autograd of these functions is the reference for the backward kernels.

All functions work in the dtype of their inputs (fp32 for golden checks, fp64 when used as the
high-precision reference for the bf16 HIP path).
"""
import math

import torch
import torch.nn.functional as F

from .geometry import (window_index_1d, window_index_2d, rpe_index_2d, t5_bucket,
                       adaptive_pool_matrix, causal_window_index_1d, t5_bucket_causal)

MASK_VAL = -5e4      # finite mask value of local/EVA (local_attention.py:141, eva.py:139)


# --------------------------------------------------------------------------------------
# argparse defaults of the reference (eva.py:235-243, lara.py:253-267, ...)
# --------------------------------------------------------------------------------------
def default_args(attn):
    base = dict(fp32=False, qkv_bias=True, attn_drop=0.0, proj_drop=0.0)
    if attn in ("local", "eva"):
        # constructor defaults (local_attention.py:27-31), not the CLI defaults
        base.update(use_rpe=False, window_size=2, attn_2d=False, overlap_window=False)
    if attn == "eva":
        base.update(adaptive_proj="default", num_landmarks=49, use_t5_rpe=False)
    if attn == "lara":
        base.update(num_landmarks=49, kernel_size=None, proposal_gen="pool",
                    use_antithetics=False, use_multisample=False, pool_module_type="light",
                    mis_type="mis-opt", alpha_coeff=1.0)
    if attn == "ra":
        base.update(num_samples=1)
    if attn == "scatterbrain":
        base.update(use_rpe=False, window_size=2, attn_2d=False, overlap_window=False,
                    approx_attn_dim=64, proj_method="favorp", cos_weighting=False, sample_scheme="default")
    if attn == "performer":
        base.update(approx_attn_dim=64, proj_method="favorp", cos_weighting=False,
                    sample_scheme="default")
    return base


# --------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------
def _gather_tokens(t, idx):
    """t [B,h,N,d], idx [G,S] (-1 = outside) -> [B,h,G,S,d] with zeros outside."""
    B, h, N, d = t.shape
    tp = torch.cat([t, t.new_zeros(B, h, 1, d)], dim=2)
    return tp[:, :, torch.where(idx < 0, torch.full_like(idx, N), idx)]


def _gather_mask(mask, idx):
    """mask [B,N] bool (True = pad), idx [G,S] -> [B,G,S] bool, True outside (pad_val=1)."""
    B, N = mask.shape
    mp = torch.cat([mask, mask.new_ones(B, 1)], dim=1)
    return mp[:, torch.where(idx < 0, torch.full_like(idx, N), idx)]


def _scatter_windows(out_w, idx_q, n):
    """out_w [B,h,G,Wq,d], idx_q [G,Wq] (a partition of [0,n)) -> [B,h,n,d]."""
    B, h, G, Wq, d = out_w.shape
    out = out_w.new_zeros(B, h, n, d)
    out[:, :, idx_q.reshape(-1)] = out_w.reshape(B, h, G * Wq, d)
    return out


def _split_heads(x, params, h):
    """qkv Linear + head split (abstract_attention.py:72-78). x [B,N,C] -> q,k,v [B,h,N,d]."""
    B, N, C = x.shape
    qkv = F.linear(x, params["qkv.weight"], params.get("qkv.bias"))
    qkv = qkv.reshape(B, N, 3, h, C // h).permute(2, 0, 3, 1, 4)
    return qkv[0], qkv[1], qkv[2]


def _merge_proj(out, params, B, seq_shape, C):
    """[B,h,N,d] -> proj([B,*seq,C]) (abstract_attention.py:86-88)."""
    x = out.transpose(1, 2).reshape((B,) + tuple(seq_shape) + (C,))
    return F.linear(x, params["proj.weight"], params["proj.bias"])


def _mlp(x, params, prefix, with_ln=True):
    """adaptive_mu_* / *_bar_gen body: Linear(d,d)[+LayerNorm(d)] (eva.py:78-98, lara.py:45-46)."""
    i_lin, i_ln = prefix
    y = F.linear(x, params[i_lin + ".weight"], params[i_lin + ".bias"])
    if with_ln:
        y = F.layer_norm(y, (y.shape[-1],), params[i_ln + ".weight"], params[i_ln + ".bias"], 1e-5)
    return y


def prm_log_features(data, proj, scale):
    """prm_projection(normalize=False) (attn_utils.py:324-336,347):
    out[..., c, n] = scale * <proj_c, data_n> - scale * |data_n|^2 / 2."""
    dash = scale * torch.einsum("...cd,...nd->...cn", proj, data)
    norm = 0.5 * scale * (data * data).sum(-1).unsqueeze(-2)
    return dash - norm


# --------------------------------------------------------------------------------------
# softmax baseline
# --------------------------------------------------------------------------------------
def softmax_core(q, k, v, mask=None, scale=None, drop_keep=None, p_drop=0.0):
    """dropout(softmax(s QK^T, -inf on padded keys)) V (abstract_attention.py:120-133); drop_keep:
    None or the 0/1 keep decisions [B,h,N,N] of attn_drop with probability p_drop."""
    scale = q.shape[-1] ** -0.5 if scale is None else scale
    s = scale * torch.einsum("bhid,bhjd->bhij", q, k)
    if mask is not None:
        s = s.masked_fill(mask.bool()[:, None, None, :], float("-inf"))
    p = torch.softmax(s, -1)
    if drop_keep is not None:
        p = p * drop_keep.reshape(p.shape).to(p.dtype) / (1.0 - p_drop)
    return torch.einsum("bhij,bhjd->bhid", p, v)


def ra_core(q, k, v, num_samples=1, noise=None, index=None, scale=None):
    """Randomized attention (randomized_attention.py:21-52).  index: the [B,h,N] draws of
    torch.multinomial(pi, 1) when num_samples is neither 0 nor -1; noise: None or [B,h,N,d]."""
    scale = q.shape[-1] ** -0.5 if scale is None else scale
    if num_samples == 0:
        mu = q + k.mean(dim=-2, keepdim=True)
    elif num_samples == -1:
        mu = q + torch.einsum("bhnm,bhmd->bhnd", torch.softmax(scale * torch.einsum("bhnd,bhmd->bhnm", q, k), -1), k)
    else:
        mu = q + torch.gather(k, 2, index.unsqueeze(-1).expand_as(q))
    w = mu if noise is None else mu + noise
    logits = scale * torch.einsum("bhnd,bhmd->bhnm", w, k) - 0.5 * scale * (k * k).sum(-1).unsqueeze(-2)
    return torch.einsum("bhnm,bhmd->bhnd", torch.softmax(logits, -1), v)


# --------------------------------------------------------------------------------------
# local window attention
# --------------------------------------------------------------------------------------
def _window_tables(attn_2d, seq_shape, n, w, e):
    """(idx_q [G,Wq], idx_k [G,Wk], n_pad).  Tokens >= n (1-D zero padding of
    pad_to_multiple, attn_utils.py:12-30) are mapped to -1 on the key side."""
    if attn_2d:
        H, W = seq_shape
        assert H % w == 0 and W % w == 0
        return window_index_2d(H, W, w, 0), window_index_2d(H, W, w, e), n
    n_pad = int(math.ceil(n / w) * w)
    idx_q = window_index_1d(n_pad, w, 0)
    idx_k = window_index_1d(n_pad, w, e)
    idx_k = torch.where(idx_k >= n, torch.full_like(idx_k, -1), idx_k)
    return idx_q, idx_k, n_pad


def local_core(q, k, v, mask, attn_2d, window_size, ext_size, bias=None, scale=None):
    """Per-window softmax(s QK^T + bias, -5e4 mask) V (local_attention.py:134-182).
    bias: None or [h, Wq, Wk] dense (table gathered by the caller)."""
    B, h, n, d = q.shape
    scale = d ** -0.5 if scale is None else scale
    if attn_2d:
        H = W = int(math.sqrt(n))
        assert H * W == n
        seq_shape = (H, W)
    else:
        seq_shape = (n,)
    idx_q, idx_k, n_pad = _window_tables(attn_2d, seq_shape, n, window_size, ext_size)
    if mask is None:
        mask = torch.zeros(B, n, dtype=torch.bool)
    mask = mask.bool()
    # queries in the 1-D pad region are dropped at the end; give them index -1 (zeros)
    idx_qg = torch.where(idx_q >= n, torch.full_like(idx_q, -1), idx_q)
    wq = _gather_tokens(q, idx_qg)
    wk = _gather_tokens(k, idx_k)
    wv = _gather_tokens(v, idx_k)
    dots = scale * torch.einsum("bhwie,bhwje->bhwij", wq, wk)
    if bias is not None:
        dots = dots + bias[None, :, None]
    wmask = _gather_mask(mask, idx_k)                    # [B,G,Wk]
    dots = dots.masked_fill(wmask[:, None, :, None, :], MASK_VAL)
    out_w = torch.einsum("bhwij,bhwje->bhwie", torch.softmax(dots, -1), wv)
    return _scatter_windows(out_w, idx_q, n_pad)[:, :, :n]


def dense_bias_2d(table, w, e):
    """[rows,h] table -> [h, w*w, (w+2e)^2] (local_attention.py:70-76)."""
    idx = rpe_index_2d(w, e)
    return table[idx.reshape(-1)].reshape(w * w, (w + 2 * e) ** 2, -1).permute(2, 0, 1)


def dense_bias_t5(emb, w, e, scale):
    """[buckets,h] embedding -> [h, Wq, Wk] * scale (eva.py:58-65,109-116); Wq/Wk are the
    flattened window sizes actually fed to it (eva.py:212-214)."""
    raise NotImplementedError  # built inline in eva_core (needs Wq, Wk)


# --------------------------------------------------------------------------------------
# EVA
# --------------------------------------------------------------------------------------
def eva_landmark_tables(attn_2d, seq_shape, n, num_landmarks, e):
    """Chunk size r and chunk index table [Cn, J] (eva.py:155-164)."""
    if attn_2d:
        H, W = seq_shape
        r = int(math.sqrt(n // num_landmarks))
        if e == 0:
            assert H % r == 0 and W % r == 0
        return r, window_index_2d(H, W, r, e)
    r = int(n // num_landmarks)
    if e == 0:
        assert n % r == 0
    return r, window_index_1d(n, r, e)


def eva_core(q, k, v, mask, attn_2d, seq_shape, window_size, ext_size, num_landmarks,
             mu_fn, noise=None, bias=None, scale=None, return_aux=False):
    """EVA's q,k,v -> out core (eva.py:145-227).

    q,k,v: [B,h,N,d] AFTER the 1-D padding of _process_input (so N is a multiple of w in
    1-D); mask: [B,N] bool or None.  mu_fn(mean_q, mean_k) -> (rf_k_bar, mu) implements the
    adaptive_proj variants (eva.py:178-185).  noise: None (eval) or [B,h,Cn,d]."""
    B, h, n, d = q.shape
    scale = d ** -0.5 if scale is None else scale
    w, e = window_size, ext_size
    if mask is None:
        mask = torch.zeros(B, n, dtype=torch.bool)
    mask = mask.bool()
    if attn_2d:
        idx_q, idx_k = window_index_2d(*seq_shape, w, 0), window_index_2d(*seq_shape, w, e)
    else:
        idx_q, idx_k = window_index_1d(n, w, 0), window_index_1d(n, w, e)
    r, idx_c = eva_landmark_tables(attn_2d, seq_shape, n, num_landmarks, e)

    # ---- landmark statistics: masked chunk means -> mu -> omega -> beta ----
    cmask = _gather_mask(mask, idx_c)                              # [B,Cn,J]
    keep = (~cmask)[:, None, :, :, None].to(q.dtype)
    cq = _gather_tokens(q, idx_c) * keep
    ck = _gather_tokens(k, idx_c) * keep
    cv = _gather_tokens(v, idx_c) * keep
    rf_k_bar, mu = mu_fn(cq.mean(-2), ck.mean(-2))
    omega = mu if noise is None else mu + noise
    logit_c = scale * torch.einsum("bhcd,bhcjd->bhcj", omega, ck) \
        - 0.5 * scale * (ck * ck).sum(-1)
    logit_c = logit_c.masked_fill(cmask[:, None], MASK_VAL)
    beta = torch.einsum("bhcj,bhcjd->bhcd", torch.softmax(logit_c, -1), cv)

    # ---- window-local logits and control-variate logits under one softmax ----
    wq = _gather_tokens(q, idx_q)
    wk = _gather_tokens(k, idx_k)
    wv = _gather_tokens(v, idx_k)
    cv_logits = scale * torch.einsum("bhwid,bhcd->bhwic", wq, rf_k_bar)
    dots = scale * torch.einsum("bhwie,bhwje->bhwij", wq, wk)
    if bias is not None:
        dots = dots + bias[None, :, None]
    wmask = _gather_mask(mask, idx_k)
    dots = dots.masked_fill(wmask[:, None, :, None, :], MASK_VAL)
    Wk = dots.shape[-1]
    p = torch.softmax(torch.cat([dots, cv_logits], -1), -1)
    out_w = torch.einsum("bhwij,bhwjd->bhwid", p[..., :Wk], wv) \
        + torch.einsum("bhwic,bhcd->bhwid", p[..., Wk:], beta)
    out = _scatter_windows(out_w, idx_q, n)
    if return_aux:
        return out, dict(beta=beta, rf_k_bar=rf_k_bar, omega=omega, mu=mu)
    return out


# --------------------------------------------------------------------------------------
# causal EVA (training / evaluation path, no incremental state)
# --------------------------------------------------------------------------------------
def causal_eva_core(q, k, v, mask, window_size, ext_size, chunk_size, mu_fn, noise=None,
                    bias=None, causal=True, scale=None, drop_keep=None, p_drop=0.0):
    """CausalEVAttention's q,k,v -> out core (causal_eva.py:666-783).

    q,k,v: [B,h,N,d] with N a multiple of the window (after _process_input); mask [B,N] bool or
    None; chunk_size r divides N.  Differences from eva_core: the window extension lies on the
    left only (:104-116), chunks are never extended (:688-694), mu = rf_q_bar + rf_k_bar (:709),
    padded QUERIES are masked as well (:742-755), and with `causal` a query sees the local keys
    up to itself (:767-773) and the control variates of the chunks before its own (:716-738).
    bias: None or [Wq, Wk] shared by the heads (:760-762).  drop_keep: None or the Bernoulli keep
    decisions [B,h,N,Wk+L] of the attention dropout with probability p_drop (:778)."""
    B, h, n, d = q.shape
    scale = d ** -0.5 if scale is None else scale
    w, e, r = window_size, ext_size, chunk_size
    assert n % w == 0 and n % r == 0 and r < n
    if mask is None:
        mask = torch.zeros(B, n, dtype=torch.bool)
    mask = mask.bool()
    idx_q = window_index_1d(n, w, 0)
    idx_k = causal_window_index_1d(n, w, e)
    idx_c = window_index_1d(n, r, 0)

    cmask = _gather_mask(mask, idx_c)                              # [B,C,r]
    keep = (~cmask)[:, None, :, :, None].to(q.dtype)
    cq = _gather_tokens(q, idx_c) * keep
    ck = _gather_tokens(k, idx_c) * keep
    cv = _gather_tokens(v, idx_c) * keep
    rf_k_bar, mu = mu_fn(cq.mean(-2), ck.mean(-2))
    omega = mu if noise is None else mu + noise
    logit_c = scale * torch.einsum("bhcd,bhcjd->bhcj", omega, ck) \
        - 0.5 * scale * (ck * ck).sum(-1)
    logit_c = logit_c.masked_fill(cmask[:, None], MASK_VAL)
    beta = torch.einsum("bhcj,bhcjd->bhcd", torch.softmax(logit_c, -1), cv)

    wq = _gather_tokens(q, idx_q)
    wk = _gather_tokens(k, idx_k)
    wv = _gather_tokens(v, idx_k)
    cv_logits = scale * torch.einsum("bhwid,bhcd->bhwic", wq, rf_k_bar)
    if causal:
        q_chunk = (idx_q // r).unsqueeze(-1)                       # [G,Wq,1]
        future = torch.arange(idx_c.shape[0]).view(1, 1, -1) >= q_chunk
        cv_logits = cv_logits.masked_fill(future[None, None], MASK_VAL)
    dots = scale * torch.einsum("bhwie,bhwje->bhwij", wq, wk)
    if bias is not None:
        dots = dots + bias
    pad = _gather_mask(mask, idx_q)[:, :, :, None] | _gather_mask(mask, idx_k)[:, :, None, :]
    dots = dots.masked_fill(pad[:, None], MASK_VAL)
    if causal:
        i = torch.arange(w).view(-1, 1)
        j = torch.arange(w + e).view(1, -1)
        dots = dots.masked_fill((j - i >= 1 + e)[None, None, None], MASK_VAL)
    Wk = dots.shape[-1]
    p = torch.softmax(torch.cat([dots, cv_logits], -1), -1)
    if drop_keep is not None:
        p = p * drop_keep.reshape(p.shape).to(p.dtype) / (1.0 - p_drop)
    out_w = torch.einsum("bhwij,bhwjd->bhwid", p[..., :Wk], wv) \
        + torch.einsum("bhwic,bhcd->bhwid", p[..., Wk:], beta)
    return _scatter_windows(out_w, idx_q, n)


# --------------------------------------------------------------------------------------
# LARA
# --------------------------------------------------------------------------------------
def lara_core(q, k, v, mask, q_bar, mu, noise=None, mis_type="mis-opt", alpha_coeff=1.0,
              sample_mode="single", scale=None, return_aux=False):
    """LinearRA estimator given landmarks (lara.py:187-246).

    q,k,v [B,h,N,d]; q_bar, mu [B,h,L,d] (mu = q_bar + k_bar, lara.py:182,185);
    noise: None (eval) | [B,h,L,d] ('single'/'antithetic') | [B,h,2L,d] ('multisample')."""
    scale = q.shape[-1] ** -0.5 if scale is None else scale
    dup = False
    if noise is None:
        omega = mu
    elif sample_mode == "multisample":
        omega, dup = mu.repeat(1, 1, 2, 1) + noise, True
    elif sample_mode == "antithetic":
        omega, dup = torch.cat([mu + noise, mu - noise], -2), True
    else:
        omega = mu + noise

    lpq = prm_log_features(q, omega, scale)                      # [B,h,C,N]
    lpk = prm_log_features(k, omega, scale)
    if mask is not None:
        lpk = lpk.masked_fill(mask.bool()[:, None, None, :], float("-inf"))
    kv_stats = torch.einsum("bhcm,bhmd->bhcd", torch.softmax(lpk, -1), v)
    lse_k = torch.logsumexp(lpk, -1, keepdim=True)

    if mis_type == "mis-biased":
        lpmu = prm_log_features(mu, omega, scale)                # [B,h,C,L]
        log_alpha = scale * torch.einsum("bhcd,bhnd->bhcn", mu, q)
        if dup:
            log_alpha = log_alpha.repeat(1, 1, 2, 1)
        log_prop = torch.logsumexp(lpmu, -1, keepdim=True)
    elif mis_type == "mis-opt":
        t = torch.softmax(scale * torch.einsum("bhcd,bhnd->bhcn", q_bar, q), -1)
        mu_c = mu
        if dup:
            mu_c, t = mu.repeat(1, 1, 2, 1), t.repeat(1, 1, 2, 1)
        lpmu = prm_log_features(mu_c, omega, scale)              # [B,h,C,C]
        log_prop = torch.diagonal(lpmu, dim1=-1, dim2=-2).unsqueeze(-1)
        bh = torch.exp(log_prop - torch.logsumexp(lpmu, -1, keepdim=True))
        alpha = bh + alpha_coeff * (t - t.mean(-2, keepdim=True))
        log_alpha = torch.log(alpha.clamp(min=1e-8))
    elif mis_type == "mis-bh":
        lpmu = prm_log_features(mu, omega, scale)
        log_alpha = 0.0
        log_prop = torch.logsumexp(lpmu, -1, keepdim=True)
    else:
        raise NotImplementedError(mis_type)

    log_iw = log_alpha + lpq + lse_k - log_prop
    sniw = torch.softmax(log_iw, -2)
    out = torch.einsum("bhcn,bhcd->bhnd", sniw, kv_stats)
    if return_aux:
        return out, dict(kv_stats=kv_stats, lse_k=lse_k, sniw=sniw, omega=omega)
    return out


def lara_landmarks_2d(q, k, v, H, W, params, args, scale):
    """_proposal_gen_2d (lara.py:129-175): adaptive 2-D average pool -> [Linear+LN] ->
    optional softmax mixing of k_bar.  Returns q_bar, k_bar [B,h,L,d]."""
    B, h, n, d = q.shape
    side = int(math.sqrt(args["num_landmarks"]))
    P = torch.kron(adaptive_pool_matrix(H, side, q.dtype), adaptive_pool_matrix(W, side, q.dtype))
    pq = torch.einsum("ln,bhnd->bhld", P, q)
    pk = torch.einsum("ln,bhnd->bhld", P, k)
    gen = args["proposal_gen"]
    if gen.startswith("pool"):
        if args["pool_module_type"] == "dense":
            # Linear/LN over all h*d channels jointly, channel order (h, d)
            def dense(p, pre):
                z = p.permute(0, 2, 1, 3).reshape(B, side * side, h * d)
                z = _mlp(z, params, (pre + ".2", pre + ".3"))
                return z.reshape(B, side * side, h, d).permute(0, 2, 1, 3)
            q_bar, k_bar = dense(pq, "q_bar_gen"), dense(pk, "k_bar_gen")
        else:
            q_bar = _mlp(pq, params, ("q_bar_gen.2", "q_bar_gen.3"))
            k_bar = _mlp(pk, params, ("k_bar_gen.2", "k_bar_gen.3"))
    elif gen.startswith("no-param-pool"):
        q_bar, k_bar = pq, pk
    else:
        raise NotImplementedError(gen)
    if gen.endswith("mixed"):
        logits = scale * torch.einsum("bhpd,bhcd->bhpc", k_bar, k_bar)
        if gen.endswith("-vmixed"):
            v_bar = torch.einsum("ln,bhnd->bhld", P, v)
            logits = logits + torch.log(v_bar.norm(dim=-1) + 1e-4).unsqueeze(-2)
        k_bar = torch.einsum("bhpc,bhcd->bhpd", torch.softmax(logits, -1), k_bar)
    return q_bar, k_bar


def lara_landmarks_1d(q, k, L, params, args):
    """Segment means of _proposal_gen_1d (lara.py:101-127); q,k already mask-zeroed."""
    B, h, n, d = q.shape
    if args["proposal_gen"].startswith("adaptive-1d"):
        q2 = _mlp(q, params, ("q_bar_gen.0", "q_bar_gen.1"))
        k2 = _mlp(k, params, ("k_bar_gen.0", "k_bar_gen.1"))
    else:
        q2, k2 = q, k
    if n <= L:
        return q2, k2
    segs = n // L
    if n % L == 0:
        sizes = [segs] * L
    else:
        num_k = (segs + 1) * L - n            # first num_k segments are short
        sizes = [segs] * num_k + [segs + 1] * (L - num_k)
    S = torch.zeros(L, n, dtype=q.dtype)
    pos = 0
    for c, sz in enumerate(sizes):
        S[c, pos:pos + sz] = 1.0 / sz
        pos += sz
    return torch.einsum("ln,bhnd->bhld", S, q2), torch.einsum("ln,bhnd->bhld", S, k2)


# --------------------------------------------------------------------------------------
# Performer (FAVOR+)
# --------------------------------------------------------------------------------------
def favorp_features(x, proj, is_query, eps=1e-4):
    """favorp_projection (kernelized_attention.py:20-56). x [B,h,N,d], proj [h,m,d]."""
    d = x.shape[-1]
    dn = d ** -0.25
    ratio = proj.shape[1] ** -0.5
    dash = torch.einsum("bhnd,hjd->bhnj", dn * x, proj)
    diag = 0.5 * dn * dn * (x * x).sum(-1, keepdim=True)
    if is_query:
        stab = dash.amax(-1, keepdim=True).detach()
    else:
        stab = dash.amax((-1, -2), keepdim=True).detach()
    return ratio * torch.exp(dash - diag - stab) + eps


def performer_core(q, k, v, mask, proj):
    """favorp features + linear attention, clamp 1e-2 (kernelized_attention.py:116-121,326-346)."""
    qp = favorp_features(q, proj, True)
    kp = favorp_features(k, proj, False)
    if mask is not None:
        kp = kp.masked_fill(mask.bool()[:, None, :, None], 0.0)
    kv = torch.einsum("bhnm,bhnd->bhmd", kp, v)
    num = torch.einsum("bhnm,bhmd->bhnd", qp, kv)
    den = torch.einsum("bhnm,bhm->bhn", qp, kp.sum(-2))
    return num / den.unsqueeze(-1).clamp(min=1e-2)


# --------------------------------------------------------------------------------------
# module-level forwards (x -> y), parameters as a dict keyed like the reference state_dict
# --------------------------------------------------------------------------------------
# --------------------------------------------------------------------------------------
# ScatterBrain (local windows + low-rank random features under one softmax)
# --------------------------------------------------------------------------------------
def scatterbrain_core(q, k, v, mask, attn_2d, seq_shape, window_size, proj, bias=None, scale=None, ext_size=0):
    """ScatterBrain's q,k,v -> out core (scatterbrain_attention.py:10-44, 71-160).  q,k,v [B,h,N,d] with N covered by whole windows; mask [B,N] bool or None; proj [h,m,d]
    random features; bias None or [h, Wq, Wk].

    log phi(x)[c] = d^-1/4 <W_c, x> - |x|^2 d^-1/2 / 2 - ln(m) / 2 (-inf for padded keys).  Per window
    g the m features act as extra softmax columns with logits log phi(q_i)[c] + log(sum over the keys
    OUTSIDE the window of phi(k_j)[c]) and values the phi-weighted mean of v over those keys."""
    B, h, n, d = q.shape
    scale = d ** -0.5 if scale is None else scale
    m = proj.shape[1]
    w = window_size
    if mask is None:
        mask = torch.zeros(B, n, dtype=torch.bool)
    mask = mask.bool()

    def log_phi(x):
        dash = d ** -0.25 * torch.einsum("bhnd,hmd->bhnm", x, proj)
        return dash - 0.5 * d ** -0.5 * (x * x).sum(-1, keepdim=True) - math.log(m) / 2
    lq = log_phi(q)
    lk = log_phi(k).masked_fill(mask[:, None, :, None], float("-inf"))
    # With window overlap (ext_size > 0) the key side of a window is the extended patch; slots outside the sequence
    # are zero padding of the reference's window_partition: v = 0 and, for the log-features, 0 as well (phi = 1, NOT
    # -inf: the partition pads with pad_val = 0, scatterbrain_attention.py:99-100), masked in the local dots only
    # (pad_val = 1, :139-144).
    e = ext_size
    if attn_2d:
        idx = window_index_2d(seq_shape[0], seq_shape[1], w, 0)
        idx_k = window_index_2d(seq_shape[0], seq_shape[1], w, e)
    else:
        idx = window_index_1d(n, w, 0)
        idx_k = window_index_1d(n, w, e)
    G, Wq = idx.shape
    flat = idx.reshape(-1)
    w_q = q[:, :, flat].reshape(B, h, G, Wq, d)
    w_k, w_v = _gather_tokens(k, idx_k), _gather_tokens(v, idx_k)
    w_lq = lq[:, :, flat].reshape(B, h, G, Wq, m)
    w_lk = _gather_tokens(lk, idx_k)
    # feature sums over all keys minus those of the window, with a shared (detached) stabiliser
    mx = torch.maximum(lk.amax(dim=-2, keepdim=True).unsqueeze(-3),
                       w_lk.amax(dim=(-2, -3), keepdim=True)).detach()             # [B,h,1,1,m]
    pk = torch.exp(lk.unsqueeze(-3) - mx)                                          # [B,h,1,N,m]
    w_pk = torch.exp(w_lk - mx)                                                    # [B,h,G,Wq,m]
    num = torch.einsum("bhtnc,bhnd->bhtcd", pk, v) - torch.einsum("bhgwc,bhgwd->bhgcd", w_pk, w_v)
    den = (pk.sum(-2) - w_pk.sum(-2)).unsqueeze(-1).clamp(min=1e-3)
    kv_stats = num / den                                                           # [B,h,G,m,d]
    lse_all = torch.logsumexp(lk.unsqueeze(-3), dim=-2, keepdim=True)              # [B,h,1,1,m]
    lse_win = torch.logsumexp(w_lk, dim=-2, keepdim=True)                          # [B,h,G,1,m]
    a = torch.maximum(lse_all, lse_win)
    nonlocal_ = a + ((lse_all - a).exp() - (lse_win - a).exp() + 1e-5).log()       # attn_utils.log_add_exp, mask (1,-1)
    log_rfa = w_lq + nonlocal_                                                     # [B,h,G,Wq,m]
    dots = scale * torch.einsum("bhwie,bhwje->bhwij", w_q, w_k)
    if bias is not None:
        dots = dots + bias[None, :, None]
    wmask = _gather_mask(mask, idx_k)[:, None, :, None, :]
    dots = dots.masked_fill(wmask, float("-inf"))
    Wk = idx_k.shape[1]
    p = torch.softmax(torch.cat([dots, log_rfa], -1), -1)
    out_w = torch.einsum("bhwij,bhwjd->bhwid", p[..., :Wk], w_v) \
        + torch.einsum("bhwic,bhwcd->bhwid", p[..., Wk:], kv_stats)
    return _scatter_windows(out_w, idx, n)


def _local_bias(params, args, h, e, scale, Wq=None, Wk=None):
    w = args["window_size"]
    if args.get("use_t5_rpe", False):
        nb = max(min(int((w + e) / 2), 64), 16)                  # eva.py:113-115
        bucket = t5_bucket(Wq, Wk, nb, w + e)
        emb = params["rel_pos_bias.relative_attention_bias.weight"]
        return emb[bucket].permute(2, 0, 1) * scale
    if args.get("use_rpe", False) and w > 0:
        tab = params["local_relative_position_bias_table"]
        if args["attn_2d"]:
            return dense_bias_2d(tab, w, e)
        return tab                                               # [h, w, w+2e]
    return None


def module_forward(attn, args, params, x, mask=None, training=False, noise_fn=None, keep_fn=None,
                   index_fn=None, qmask_fn=None):
    """y = module(x, key_padding_mask) for attn in {softmax, local, eva, lara, performer,
    causal_eva (batch-first x; training/evaluation path)}.

    args: constructor kwargs (missing ones take default_args); params: dict of tensors with
    the reference's state_dict keys; noise_fn(shape) -> standard-normal tensor for the i-th
    sampling call of a training-mode forward; keep_fn(shape) -> 0/1 keep decisions of an attention
    dropout; index_fn(shape) -> the key indices randomized attention draws; qmask_fn(n) -> the 0/1 block-drop
    decisions of a quantization-noise draw (causal EVA with q_noise > 0)."""
    if attn == "causal_eva":
        return _causal_eva_forward(args, params, x, mask, training, noise_fn, keep_fn, qmask_fn)
    a = default_args(attn)
    a.update(args)
    h = a["num_heads"]
    B, *seq_shape, C = x.shape
    d = C // h
    scale = d ** -0.5

    if attn == "softmax":
        n = int(math.prod(seq_shape))
        q, k, v = _split_heads(x.reshape(B, n, C), params, h)
        p_drop = float(a["attn_drop"])
        keep = keep_fn((B, h, n, n)) if (training and p_drop > 0) else None
        return _merge_proj(softmax_core(q, k, v, mask, scale, keep, p_drop), params, B, seq_shape, C)

    if attn == "scatterbrain":
        w = a["window_size"]
        e = max(1, w // 2) if a["overlap_window"] else 0
        orig_n = int(math.prod(seq_shape))
        if a["attn_2d"]:
            n, xs = orig_n, x.reshape(B, orig_n, C)
        else:
            n = int(math.ceil(orig_n / w) * w)                   # _process_input: pad x, extend the mask
            xs = F.pad(x, (0, 0, 0, n - orig_n))
            pad_mask = torch.zeros(B, n, dtype=torch.bool)
            pad_mask[:, orig_n:] = True
            if mask is not None:
                pad_mask[:, :orig_n] = mask.bool()
            mask = pad_mask
            seq_shape = [n]
        q, k, v = _split_heads(xs, params, h)
        proj = noise_fn((h, a["approx_attn_dim"], d)).to(x.dtype) if training else params["eval_proj"]
        bias = _local_bias(params, a, h, e, scale)
        out = scatterbrain_core(q, k, v, mask, a["attn_2d"], seq_shape, w, proj, bias, scale, ext_size=e)
        y = F.linear(out.permute(0, 2, 1, 3).reshape((B,) + tuple(seq_shape) + (C,)),
                     params["proj.weight"], params["proj.bias"])
        return y if a["attn_2d"] else y[..., :orig_n, :]

    if attn == "ra":
        n = int(math.prod(seq_shape))
        q, k, v = _split_heads(x.reshape(B, n, C), params, h)
        ns = a["num_samples"]
        index = index_fn((B, h, n)) if ns not in (0, -1) else None
        noise = noise_fn((B, h, n, d)).to(x.dtype) if training else None
        return _merge_proj(ra_core(q, k, v, ns, noise, index, scale), params, B, seq_shape, C)

    if attn == "performer":
        n = int(math.prod(seq_shape))
        q, k, v = _split_heads(x.reshape(B, n, C), params, h)
        if training:
            proj = noise_fn((h, a["approx_attn_dim"], d)).to(x.dtype)
        else:
            proj = params["eval_proj"]
        return _merge_proj(performer_core(q, k, v, mask, proj), params, B, seq_shape, C)

    if attn == "local":
        n = int(math.prod(seq_shape))
        w = a["window_size"]
        e = max(1, w // 2) if a["overlap_window"] else 0
        q, k, v = _split_heads(x.reshape(B, n, C), params, h)
        bias = _local_bias(params, a, h, e, scale)
        out = local_core(q, k, v, mask, a["attn_2d"], w, e, bias, scale)
        return _merge_proj(out, params, B, seq_shape, C)

    if attn == "eva":
        w = a["window_size"]
        e = max(1, w // 2) if a["overlap_window"] else 0
        orig_n = int(math.prod(seq_shape))
        if a["attn_2d"]:
            assert len(seq_shape) == 2 and seq_shape[0] % w == 0 and seq_shape[1] % w == 0
            n = orig_n
            xs = x.reshape(B, n, C)
        else:
            # _process_input (eva.py:127-136): pad x (not q/k/v) and synthesise/extend the mask
            n = int(math.ceil(orig_n / w) * w)
            xs = F.pad(x, (0, 0, 0, n - orig_n))
            pad_mask = torch.zeros(B, n, dtype=torch.bool)
            pad_mask[:, orig_n:] = True
            if mask is not None:
                pad_mask[:, :orig_n] = mask.bool()
            mask = pad_mask
            seq_shape = [n]
        q, k, v = _split_heads(xs, params, h)
        ap = a["adaptive_proj"]

        def mu_fn(mq, mk):
            if ap in ("default", "no-ln"):
                ln = ap == "default"
                rq = _mlp(mq, params, ("adaptive_mu_q.0", "adaptive_mu_q.1"), ln)
                rk = _mlp(mk, params, ("adaptive_mu_k.0", "adaptive_mu_k.1"), ln)
                return rk, 0.5 * (rq + rk)
            rk = _mlp(mk, params, ("adaptive_mu_k.0", "adaptive_mu_k.1"), True)
            return rk, torch.zeros_like(rk)

        Wq = w * w if a["attn_2d"] else w
        Wk = (w + 2 * e) ** 2 if a["attn_2d"] else w + 2 * e
        bias = _local_bias(params, a, h, e, scale, Wq, Wk)
        _, idx_c = eva_landmark_tables(a["attn_2d"], seq_shape, n, a["num_landmarks"], e)
        noise = noise_fn((B, h, idx_c.shape[0], d)).to(x.dtype) if training else None
        out = eva_core(q, k, v, mask, a["attn_2d"], seq_shape, w, e, a["num_landmarks"],
                       mu_fn, noise, bias, scale)
        y = F.linear(out.permute(0, 2, 1, 3).reshape((B,) + tuple(seq_shape) + (C,)),
                     params["proj.weight"], params["proj.bias"])
        if not a["attn_2d"]:
            y = y[..., :orig_n, :]
        return y

    if attn == "lara":
        L = a["num_landmarks"]
        n = int(math.prod(seq_shape))
        q, k, v = _split_heads(x.reshape(B, n, C), params, h)
        if len(seq_shape) == 2:
            q_bar, k_bar = lara_landmarks_2d(q, k, v, seq_shape[0], seq_shape[1], params, a, scale)
        else:
            if mask is not None:
                keep = (~mask.bool())[:, None, :, None].to(x.dtype)
                q, k, v = q * keep, k * keep, v * keep
            q_bar, k_bar = lara_landmarks_1d(q, k, L, params, a)
        mu = q_bar + k_bar
        mode, noise = "single", None
        if training:
            if a["use_multisample"]:
                mode = "multisample"
                noise = noise_fn((B, h, mu.shape[-2] * 2, d)).to(x.dtype)
            elif a["use_antithetics"]:
                mode = "antithetic"
                noise = noise_fn(tuple(mu.shape)).to(x.dtype)
            else:
                noise = noise_fn(tuple(mu.shape)).to(x.dtype)
        out = lara_core(q, k, v, mask, q_bar, mu, noise, a["mis_type"], a["alpha_coeff"],
                        mode, scale)
        return _merge_proj(out, params, B, seq_shape, C)

    raise KeyError(attn)


def quant_noise_(params, name, p, block, qmask_fn):
    """Quantization noise on one projection, as the forward pre-hook of causal_eva.py:165-213 applies it in training mode:
    every run of `block` consecutive INPUT features of every output row is dropped with probability p and the survivors are
    scaled by 1 / (1 - p).  The reference writes the result through `weight.data`, i.e. the parameter itself changes and the
    gradient it receives is the one with respect to the noised values (dropped blocks get gradient too): so does this."""
    w = params[name + ".weight"]
    out_f, in_f = w.shape
    assert in_f % block == 0, "Input features must be a multiple of block sizes"      # :149-151
    m = qmask_fn(in_f // block * out_f).to(torch.float32).reshape(-1)
    m = m.repeat_interleave(block, -1).view(-1, in_f).to(torch.bool)
    w.data = (1.0 / (1.0 - p)) * w.data.masked_fill(m, 0)


def _causal_eva_forward(args, params, x, mask, training, noise_fn, keep_fn=None, qmask_fn=None):
    """CausalEVAttention.forward without incremental state (causal_eva.py:458-536,666-790) on
    batch-first x [B,T,C] (the module itself is time-first; callers transpose).  args: the
    constructor kwargs with `attn_args` as a dict (window_size, overlap_window, causal,
    num_chunks, chunk_size, use_t5_rpe, adaptive_proj) and optionally q_noise / qn_block_size
    (:339-351; the hooks fire in call order q, k, v (:511-513), then out (:785))."""
    aa = dict(adaptive_proj="default", num_chunks=None, chunk_size=None, causal=False,
              use_t5_rpe=False, window_size=4, overlap_window=False)
    aa.update(args["attn_args"])
    h = args["num_heads"]
    B, T, C = x.shape
    d = C // h
    scale = d ** -0.5
    w = aa["window_size"]
    e = max(1, w) if aa["overlap_window"] else 0                 # :354-357 (not w // 2)
    n = int(math.ceil(T / w) * w)
    xs = F.pad(x, (0, 0, 0, n - T))
    pad_mask = torch.zeros(B, n, dtype=torch.bool)
    pad_mask[:, T:] = True
    if mask is not None:
        pad_mask[:, :T] = mask.bool()

    qn = float(args.get("q_noise", 0.0)) if training else 0.0
    qn_block = int(args.get("qn_block_size", 8))

    def heads(name):
        if qn > 0:
            quant_noise_(params, name, qn, qn_block, qmask_fn)
        y = F.linear(xs, params[name + ".weight"], params.get(name + ".bias"))
        return y.reshape(B, n, h, d).transpose(1, 2)
    q, k, v = heads("q_proj"), heads("k_proj"), heads("v_proj")
    r = aa["chunk_size"] if aa["chunk_size"] is not None else int(n // aa["num_chunks"])
    ap = aa["adaptive_proj"]
    assert ap in ("qk", "no-ln")

    def mu_fn(mq, mk):
        rq = _mlp(mq, params, ("adaptive_mu_q.0", "adaptive_mu_q.1"), ap == "qk")
        rk = _mlp(mk, params, ("adaptive_mu_k.0", "adaptive_mu_k.1"), ap == "qk")
        return rk, rq + rk

    bias = None
    if aa["use_t5_rpe"] and w > 0:
        nb = max(min(int((w + e) / 2), 64), 16)                  # :368-374
        bucket = (t5_bucket_causal if aa["causal"] else t5_bucket)(w, w + e, nb, w + e)
        bias = params["rel_pos_bias.relative_attention_bias.weight"][bucket][..., 0] * scale
    noise = noise_fn((B, h, n // r, d)).to(x.dtype) if training else None
    p_drop = float(args.get("dropout", 0.0))
    keep = keep_fn((B, h, n, w + e + n // r)) if (training and p_drop > 0) else None
    out = causal_eva_core(q, k, v, pad_mask, w, e, r, mu_fn, noise, bias, aa["causal"], scale,
                          keep, p_drop)
    if qn > 0:
        quant_noise_(params, "out_proj", qn, qn_block, qmask_fn)
    y = F.linear(out.transpose(1, 2).reshape(B, n, C), params["out_proj.weight"],
                 params.get("out_proj.bias"))
    return y[:, :T]
