"""Softmax baseline and the base class of every attention module.

Mirrors efficient_attention/abstract_attention.py:41-140 of the reference: same constructor
kwargs (`fp32` is accepted and, like the reference, never read), same parameters
(`qkv`, `proj`), same init (trunc-normal std .02 Linear weights, zero biases, unit LayerNorm),
same `forward(x, key_padding_mask=None)` protocol with `x: [B, *seq_shape, C]`.
"""
import math

import torch
import torch.nn as nn

from . import add_nested_argument
from . import _ops
from . import _f32


class AbstractAttention(nn.Module):
    """Unused stub kept for import compatibility (reference :10-39)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        self.name = "%s.%d" % (self.__class__.__name__, hash(self))

    def _reset_parameters(self):
        raise NotImplementedError

    def forward(self, *args, **kwargs):
        raise NotImplementedError

    def _apply_attention(self, *args, **kwargs):
        raise NotImplementedError


class MultiheadAttention(nn.Module):
    # fp32 activations outside autocast stay fp32 and take the fp32-faithful kernels (_f32.py): the softmax baseline,
    # LocalAttention and EVA; subclasses whose cores only exist with 16-bit operands set this to False
    _F32_CORE = True

    def __init__(self, dim, num_heads, fp32=False, qkv_bias=True, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.dim = dim
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv_bias = qkv_bias
        self.fp32 = fp32
        self.qkv = nn.Linear(dim, 3 * dim, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self._keep_mask_fn = None                      # tests: injected attention-dropout decisions
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.LayerNorm):
            nn.init.zeros_(m.bias)
            nn.init.ones_(m.weight)
        elif isinstance(m, nn.Conv2d):
            fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
            m.weight.data.normal_(0, math.sqrt(2.0 / fan_out))
            if m.bias is not None:
                m.bias.data.zero_()

    # ---- projections ---------------------------------------------------------------------
    def project_qkv(self, x):
        """x [B, N, C] -> fused qkv [B, N, 3, h, d] in the kernels' I/O dtype (bf16/fp16).
        Under autocast the Linear already emits it; fp32 activations are rounded to bf16 (the
        kernels compute bf16 x bf16 -> fp32, the autocast contract of vit/engine.py:47)."""
        B, N, C = x.shape
        qkv = _ops.linear(x, self.qkv)
        if self._F32_CORE and _f32.usable(qkv) and self.head_dim in (32, 64, 128):
            # round 5: fp32 activations outside autocast stay fp32 -- the core then runs on the fp32-faithful kernels
            # (ea_f32_attn_*), as the reference computes there (abstract_attention.py:120-133)
            return qkv.reshape(B, N, 3, self.num_heads, C // self.num_heads)
        qkv = _ops.to_io_dtype(qkv)
        return qkv.reshape(B, N, 3, self.num_heads, C // self.num_heads)

    project_qkv._ea_builtin = True

    def proj_and_split_heads(self, x):
        """Reference-compatible helper (:72-78): three [B, h, N, d] strided views."""
        B, *seq_shape, C = x.shape
        N = int(math.prod(seq_shape))
        return _ops._qkv_views(self.project_qkv(x.reshape(B, N, C)))

    def merge_and_project(self, out, B, seq_shape, C, dtype):
        """out [B, N, h, d] (contiguous) -> proj -> proj_drop, shaped [B, *seq_shape, C]."""
        x = out.reshape((B,) + tuple(seq_shape) + (C,))
        x = _ops.linear(x, self.proj)
        if not torch.is_autocast_enabled() and x.dtype != dtype:
            x = x.to(dtype)
        return self.proj_drop(x)

    # ---- softmax attention ---------------------------------------------------------------
    def forward(self, x, key_padding_mask=None):
        B, *seq_shape, C = x.shape
        N = int(math.prod(seq_shape))
        # the common training case as ONE autograd node (projections + core, round 4): decided before anything is launched
        # (a subclass that overrides project_qkv / merge_and_project -- q/k norms, LoRA on qkv ... -- keeps the three-node path
        #  that calls them: ADVICE r04)
        if (torch.is_autocast_enabled() and x.is_cuda and (self.proj_drop.p == 0.0 or not self.training)
                and getattr(type(self).project_qkv, "_ea_builtin", False)
                and type(self).merge_and_project is MultiheadAttention.merge_and_project
                and _ops.core_module_fn_supported(x, self.qkv, self.proj, torch.get_autocast_dtype("cuda"))):
            core, inputs = self._core_spec(B, N, seq_shape, key_padding_mask, x.device)
            if core is not None:
                y = _ops.CoreModuleFn.apply(x, self.qkv.weight, self.qkv.bias, self.proj.weight, self.proj.bias, core,
                                            torch.get_autocast_dtype("cuda"), self.num_heads, *inputs)
                return self.proj_drop(y)
        qkv5 = self.project_qkv(x.reshape(B, N, C))
        out = self._attend(qkv5, key_padding_mask, seq_shape)        # [B, N, h, d]
        return self.merge_and_project(out, B, seq_shape, C, x.dtype)

    def _core_spec(self, B, N, seq_shape, key_padding_mask, device):
        """(core spec for _ops.CoreModuleFn, its differentiable inputs), or (None, ()) for subclasses that only override
        _attend: they keep the three-node path."""
        if type(self)._attend is not MultiheadAttention._attend:
            return self._graph_core_spec(key_padding_mask, seq_shape)
        mask = _ops._mask_u8(key_padding_mask, B, N, device)
        return _ops.SoftmaxCore(mask, *self._attn_keep(B, N, device)), ()

    # A subclass whose `_attend` is a chain of autograd Functions of its own (randomized attention) joins the single-node path by
    # NAMING the parameters that chain reads besides qkv / proj (round 6, _ops.GraphCore: the chain is recorded inside the node's
    # forward and differentiated inside its backward; a parameter it reads but does not name would get no gradient, hence opt-in)
    _graph_core_params = None            # None: keep the three-node path; a tuple of attribute names (may be empty) opts in

    def _graph_core_spec(self, key_padding_mask, seq_shape):
        names = type(self)._graph_core_params
        if names is None or not _ops.USE_GRAPH_CORE:
            return None, ()
        params = tuple(getattr(self, n) for n in names)
        need = torch.is_grad_enabled()
        return _ops.GraphCore(lambda qkv5, *ps: self._attend(qkv5, key_padding_mask, seq_shape), len(params), need), params

    def _attend(self, qkv5, key_padding_mask, seq_shape):
        B, N = qkv5.shape[:2]
        mask = _ops._mask_u8(key_padding_mask, B, N, qkv5.device)
        if qkv5.dtype == torch.float32:
            return _f32.softmax_core(qkv5, mask, *self._attn_keep(B, N, qkv5.device))
        return _ops.SoftmaxAttnFn.apply(qkv5, mask, *self._attn_keep(B, N, qkv5.device))

    def _attn_keep(self, B, N, device):
        """Attention dropout (reference :131, `attn = self.attn_drop(attn)` on [B,h,N,N]): the Bernoulli
        keep decisions are drawn here, one per (query, key), and applied inside the kernels."""
        p = self.attn_drop.p
        if not (self.training and p > 0):
            return ()
        if p >= 1:
            raise NotImplementedError("attention dropout with p = 1")
        h = self.num_heads
        ld = -(-N // 64) * 64
        if self._keep_mask_fn is not None:
            keep = self._keep_mask_fn((B, h, N, N)).to(device=device, dtype=torch.uint8)
            if ld != N:
                keep = torch.nn.functional.pad(keep, (0, ld - N))
        else:
            keep = torch.empty((B, h, N, ld), device=device, dtype=torch.uint8).bernoulli_(1 - p)
        return (keep.contiguous(), 1.0 / (1.0 - p))

    @staticmethod
    def add_attn_specific_args(parent_parser, struct_name="attn_args", prefix=""):
        group = parent_parser.add_argument_group("Attention")
        flag_prefix = prefix + "-" if len(prefix) > 1 else ""
        add_nested_argument(group, "--%sfp32" % flag_prefix, struct_name=struct_name, prefix=prefix,
                            default=False, action="store_true")
        return parent_parser
