"""Index geometry of the hot path (test infrastructure; see oracle/__init__.py).

The reference builds windows with F.pad + as_strided views and einops rearranges
(attn_utils.py:152-234).  Here the same token<->(window, slot) maps are written as explicit
integer index tables (-1 = "outside the sequence/grid", i.e. the reference's pad value), so
every core below is a plain gather -- a different construction of the same map.
"""
import math
import torch


def window_index_1d(n, w, e=0):
    """[n//w, w+2e] token index of slot j of window g, -1 where the extended window leaves
    [0, n).  Restates window_1d_partition (attn_utils.py:155-166): pad e each side, window g
    starts at padded position g*w."""
    g = torch.arange(n // w).unsqueeze(1)
    j = torch.arange(w + 2 * e).unsqueeze(0)
    tok = g * w - e + j
    return torch.where((tok >= 0) & (tok < n), tok, torch.full_like(tok, -1))


def window_index_2d(H, W, w, e=0):
    """[(H//w)*(W//w), (w+2e)^2] token index (row-major y*W+x), -1 outside the grid.
    Restates window_2d_partition (attn_utils.py:190-210): windows ordered (h1, w1) row-major,
    slots ordered (i, j) row-major over the extended (w+2e)x(w+2e) patch."""
    t = w + 2 * e
    h1 = torch.arange(H // w).view(-1, 1, 1, 1)
    w1 = torch.arange(W // w).view(1, -1, 1, 1)
    i = torch.arange(t).view(1, 1, -1, 1)
    j = torch.arange(t).view(1, 1, 1, -1)
    y = h1 * w - e + i
    x = w1 * w - e + j
    ok = (y >= 0) & (y < H) & (x >= 0) & (x < W)
    tok = torch.where(ok, y * W + x, torch.full_like(y * W + x, -1))
    return tok.reshape((H // w) * (W // w), t * t)


def rpe_index_2d(w, e=0):
    """[w*w, (w+2e)^2] index into the learned 2-D relative-position table.
    Restates local_attention.py:49-62: with q=(qi,qj) in [0,w)^2 and k=(ki,kj) in
    [-e, w+e)^2, index = (qi-ki+e+w-1)*(2e+w) + (qj-kj+e+w-1).  (The row multiplier is 2e+w,
    so distinct offsets can collide -- reproduced as is.)"""
    q = torch.arange(w)
    k = torch.arange(-e, w + e)
    qi, qj = q.view(-1, 1, 1, 1), q.view(1, -1, 1, 1)
    ki, kj = k.view(1, 1, -1, 1), k.view(1, 1, 1, -1)
    shift = e + w - 1
    idx = (qi - ki + shift) * (2 * e + w) + (qj - kj + shift)
    return idx.reshape(w * w, (w + 2 * e) ** 2)


def rpe_table_rows_2d(w, e=0):
    """Number of rows of the 2-D table (local_attention.py:46-47)."""
    return 2 * (w + e - 1) * (2 * e + w + 1) + 1


def t5_bucket(i_len, j_len, num_buckets, max_distance):
    """[i_len, j_len] bucket of (k_pos - q_pos), non-causal T5 scheme.
    Restates T5RelativePositionBias._relative_position_bucket / forward (eva.py:31-64)."""
    q_pos = torch.arange(i_len).view(-1, 1)
    k_pos = torch.arange(j_len).view(1, -1)
    n = q_pos - k_pos                    # = -(k_pos - q_pos)
    nb = num_buckets // 2
    ret = (n < 0).long() * nb
    n = n.abs()
    max_exact = nb // 2
    is_small = n < max_exact
    nf = n.clamp(min=1).double()
    large = max_exact + (torch.log(nf.float() / max_exact)
                         / math.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.minimum(large, torch.full_like(large, nb - 1))
    return ret + torch.where(is_small, n, large)


def causal_window_index_1d(n, w, e=0):
    """[n//w, e+w] token index of slot j of window g with the extension on the LEFT only, -1 before
    the sequence start.  Restates causal_window_1d_partition (causal_eva.py:104-116): pad e in
    front, window g starts at padded position g*w."""
    g = torch.arange(n // w).unsqueeze(1)
    j = torch.arange(e + w).unsqueeze(0)
    tok = g * w - e + j
    return torch.where(tok >= 0, tok, torch.full_like(tok, -1))


def t5_bucket_causal(i_len, j_len, num_buckets, max_distance):
    """[i_len, j_len] bucket of (k_pos - q_pos) in the causal T5 scheme: distances into the future
    collapse to 0, no sign split (causal_eva.py:62-99 with causal=True)."""
    q_pos = torch.arange(i_len).view(-1, 1)
    k_pos = torch.arange(j_len).view(1, -1)
    n = (q_pos - k_pos).clamp(min=0)
    max_exact = num_buckets // 2
    nf = n.clamp(min=1).float()
    large = max_exact + (torch.log(nf / max_exact)
                         / math.log(max_distance / max_exact) * (num_buckets - max_exact)).long()
    large = torch.minimum(large, torch.full_like(large, num_buckets - 1))
    return torch.where(n < max_exact, n, large)


def adaptive_pool_matrix(in_size, out_size, dtype=torch.float32):
    """[out_size, in_size] averaging matrix of nn.AdaptiveAvgPool1d: bin o covers
    [floor(o*in/out), ceil((o+1)*in/out)).  The 2-D pool of lara.py:43,48 is the Kronecker
    product of two of these."""
    P = torch.zeros(out_size, in_size, dtype=dtype)
    for o in range(out_size):
        s = (o * in_size) // out_size
        t = -((-(o + 1) * in_size) // out_size)
        P[o, s:t] = 1.0 / (t - s)
    return P
