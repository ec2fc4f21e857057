"""Local (windowed) attention baseline, and the base class of EVA.

Mirrors efficient_attention/local_attention.py:25-194: constructor kwargs, the learned
relative-position tables (2-D table + `relative_position_index` buffer, 1-D table), the
overlap rule `ext_size = max(1, window_size // 2)`, and the argparse flags.  The per-window
softmax(s QK^T + bias, -5e4 mask) V runs in libea_hip.so (ea_window_attn_fwd/bwd with L = 0);
windows are never materialised.
"""
import math

import torch
import torch.nn as nn

from . import add_nested_argument
from . import _ops
from . import _f32
from .abstract_attention import MultiheadAttention


class _TableGather(torch.autograd.Function):
    """table[idx] (reference local_attention.py:70-79 `relative_position_bias_table[index]`) whose backward adds, for every
    table row, the gradients of the positions that read it in a fixed order (ea_gather_sum; `inv` [rows, K] lists those
    positions) instead of autograd's sort-based index_put (six kernels for a 169 x 3 table): deterministic, one launch."""

    @staticmethod
    def forward(ctx, table, idx, inv):
        ctx.save_for_backward(inv)
        return table.index_select(0, idx)

    @staticmethod
    def backward(ctx, g):
        (inv,) = ctx.saved_tensors
        if not g.is_cuda:                                     # (CPU construction-time checks only: no kernel there)
            out = torch.zeros((inv.shape[0],) + tuple(g.shape[1:]), dtype=torch.float32)
            gz = torch.cat([g.float(), torch.zeros((1,) + tuple(g.shape[1:]))])
            return gz[torch.where(inv < 0, torch.full_like(inv, g.shape[0]), inv).long()].sum(1).to(g.dtype), None, None
        g32 = g.float().contiguous()
        cols = g32[0].numel()
        out = torch.empty((inv.shape[0], cols), dtype=torch.float32, device=g.device)
        _ops.nv.call("ea_gather_sum", inv.shape[0], inv.shape[1], cols, _ops.nv.ptr(g32), _ops.nv.ptr(inv), _ops.nv.ptr(out),
                     _ops.nv.stream())
        return out.view((inv.shape[0],) + tuple(g.shape[1:])).to(g.dtype), None, None


def _inverse_index(flat, rows):
    """[rows, K] int32: the positions of `flat` that hold each value in [0, rows), padded with -1."""
    order = torch.argsort(flat, stable=True)
    counts = torch.bincount(flat, minlength=rows)
    K = int(counts.max())
    inv = torch.full((rows, K), -1, dtype=torch.int32)
    start = torch.cumsum(counts, 0) - counts
    pos = torch.arange(flat.numel()) - start[flat[order]]
    inv[flat[order], pos] = order.to(torch.int32)
    return inv


def relative_position_index_2d(window_size, ext_size):
    """[w*w, (w+2e)^2] int64 index into the 2-D bias table for query (qi,qj) in [0,w)^2 and key
    (ki,kj) in [-e, w+e)^2: (qi-ki+e+w-1)*(2e+w) + (qj-kj+e+w-1)  (reference :49-62)."""
    w, e = window_size, ext_size
    q = torch.arange(w)
    k = torch.arange(-e, w + e)
    off = e + w - 1
    rows = (q.view(w, 1, 1, 1) - k.view(1, 1, -1, 1) + off) * (2 * e + w)
    cols = q.view(1, w, 1, 1) - k.view(1, 1, 1, -1) + off
    return (rows + cols).reshape(w * w, (w + 2 * e) ** 2)


class LocalAttention(MultiheadAttention):
    def __init__(self, use_rpe=False, window_size=2, attn_2d=False, overlap_window=False,
                 *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.window_size = window_size
        self.attn_2d = attn_2d
        self.use_rpe = use_rpe if window_size > 0 else False
        self.ext_size = max(1, window_size // 2) if overlap_window else 0
        if self.use_rpe:
            w, e = window_size, self.ext_size
            if attn_2d:
                rows = 2 * (w + e - 1) * (2 * e + w + 1) + 1
                self.local_relative_position_bias_table = nn.Parameter(torch.zeros(rows, self.num_heads))
                self.register_buffer("relative_position_index", relative_position_index_2d(w, e))
                flat = self.relative_position_index.reshape(-1)
                self.register_buffer("_rpe_inv", _inverse_index(flat, rows), persistent=False)   # not part of state_dict
            else:
                self.local_relative_position_bias_table = nn.Parameter(
                    torch.zeros(self.num_heads, w, w + 2 * e))
            nn.init.trunc_normal_(self.local_relative_position_bias_table, std=.02)
        self.apply(self._init_weights)

    # ---- dense per-head bias [h, Wq, Wk] handed to the kernel ---------------------------
    def _table_bias(self):
        if not self.use_rpe:
            return None
        tab = self.local_relative_position_bias_table
        if not self.attn_2d:
            return tab
        idx = self.relative_position_index
        return _TableGather.apply(tab, idx.reshape(-1), self._rpe_inv).reshape(
            idx.shape[0], idx.shape[1], -1).permute(2, 0, 1)

    def _table_spec(self):
        """(table parameter, _ops.TableBias) for the single-node module paths, which build the dense bias from the table in one
        launch each way (round 6), or (None, None): no 2-D table, EA_TABLE_BIAS=0."""
        if not (self.use_rpe and self.attn_2d and _ops.USE_TABLE_BIAS):
            return None, None
        tb = self.__dict__.get("_tb")
        if tb is None:
            idx = self.relative_position_index
            tb = _ops.TableBias(idx, self.local_relative_position_bias_table.shape[0], idx.shape[0], idx.shape[1], 1.0,
                                inv=self._rpe_inv)
            self.__dict__["_tb"] = tb
        return self.local_relative_position_bias_table, tb

    def add_rel_pos_bias(self, local_dots):
        """Reference-compatible helper (:70-79): local_dots [b,h,w,i,j] + bias."""
        return local_dots + self._table_bias().unsqueeze(0).unsqueeze(2)

    def _geometry(self, N, seq_shape):
        if self.attn_2d:
            if len(seq_shape) == 2:
                H, W = seq_shape
            else:
                H = W = int(math.sqrt(N))              # reference :142-146
            assert H * W == N
            return True, (H, W)
        return False, (N,)

    def _attend(self, qkv5, key_padding_mask, seq_shape):
        B, N = qkv5.shape[:2]
        attn_2d, shape = self._geometry(N, seq_shape)
        if attn_2d:
            H = W = int(math.sqrt(N))
            assert H * W == N, "LocalAttention with attn_2d expects a square grid"
            assert H % self.window_size == 0
            shape = (H, W)
        mask = _ops._mask_u8(key_padding_mask, B, N, qkv5.device)
        if qkv5.dtype == torch.float32:
            return _f32.local_core(qkv5, self._table_bias(), mask, attn_2d, shape, self.window_size, self.ext_size)
        return _ops.LocalAttnFn.apply(qkv5, self._table_bias(), mask, attn_2d, shape,
                                      self.window_size, self.ext_size)

    def _core_spec(self, B, N, seq_shape, key_padding_mask, device):
        if type(self)._attend is not LocalAttention._attend:
            return None, ()
        attn_2d, shape = self._geometry(N, seq_shape)
        if attn_2d:
            H = W = int(math.sqrt(N))
            assert H * W == N, "LocalAttention with attn_2d expects a square grid"
            assert H % self.window_size == 0
            shape = (H, W)
        mask = _ops._mask_u8(key_padding_mask, B, N, device)
        table, tb = self._table_spec()
        if tb is not None and device.type == "cuda":
            return _ops.LocalCore(mask, attn_2d, shape, self.window_size, self.ext_size, tb=tb), (table,)
        return _ops.LocalCore(mask, attn_2d, shape, self.window_size, self.ext_size), (self._table_bias(),)

    @staticmethod
    def add_attn_specific_args(parent_parser, struct_name="attn_args", prefix=""):
        parent_parser = MultiheadAttention.add_attn_specific_args(parent_parser, struct_name=struct_name, prefix=prefix)
        group = parent_parser.add_argument_group("Attention")
        fp = prefix + "-" if len(prefix) > 1 else ""
        kw = dict(struct_name=struct_name, prefix=prefix)
        add_nested_argument(group, "--%suse-rpe" % fp, action="store_true", default=False, **kw)
        add_nested_argument(group, "--%swindow-size" % fp, default=4, type=int, **kw)
        add_nested_argument(group, "--%sattn-2d" % fp, action="store_true", default=False, **kw)
        add_nested_argument(group, "--%soverlap-window" % fp, action="store_true", default=False, **kw)
        return parent_parser
