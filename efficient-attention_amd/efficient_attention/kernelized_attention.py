"""Performer (FAVOR+) baseline, MI355X build.

Mirrors the in-scope part of efficient_attention/kernelized_attention.py:223-359
(`proj_method='favorp'`, `sample_scheme='default'`): constructor kwargs, the `eval_proj`
buffer of per-head orthogonal random features (fresh Gaussian features on every training
call), and the argparse flags.  phi(q), phi(k), phi(K)^T V and the normalised read-out run in
libea_hip.so (_ops.PerformerAttnFn).  Other feature maps of the reference
(fourier / relu / dpfp / mlp-fourier / cos-weighting) and learnable features
(`sample_scheme='learnable'`) are outside this build's scope and raise NotImplementedError.
"""
import math

import torch
import torch.nn as nn

from . import add_nested_argument
from . import _ops
from .abstract_attention import MultiheadAttention


def gaussian_orthogonal_random_matrix(nb_rows, nb_columns, seed=0, device=None, dtype=None):
    """Stacked QR blocks with chi-distributed row norms (reference :203-221)."""
    blocks = []
    for _ in range(nb_rows // nb_columns):
        q, _ = torch.linalg.qr(torch.randn(nb_columns, nb_columns), mode='reduced')
        blocks.append(q.t())
    rest = nb_rows - (nb_rows // nb_columns) * nb_columns
    if rest > 0:
        q, _ = torch.linalg.qr(torch.randn(nb_columns, nb_columns), mode='reduced')
        blocks.append(q.t()[:rest])
    mat = torch.cat(blocks).to(device=device, dtype=dtype)
    norms = torch.randn(nb_rows, nb_columns, device=device, dtype=dtype).norm(dim=1)
    return norms.unsqueeze(1) * mat


def create_proj_matrix(num_heads, proj_dim, input_dim, ortho=False, seed=0, device=None, dtype=None):
    if not ortho:
        return torch.randn(num_heads, proj_dim, input_dim, device=device, dtype=dtype)
    return torch.stack([gaussian_orthogonal_random_matrix(proj_dim, input_dim, seed=seed + 1000 * i,
                                                          device=device, dtype=dtype)
                        for i in range(num_heads)], dim=0)


class KernelizedAttention(MultiheadAttention):
    _F32_CORE = False           # no fp32-operand kernels for this core's estimator (the Performer core has its own fp32 path)
    def __init__(self, approx_attn_dim=64, proj_method='favorp', cos_weighting=False,
                 sample_scheme='default', *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.approx_attn_dim = approx_attn_dim
        self.proj_method = proj_method
        self.cos_weighting = cos_weighting
        self.sample_scheme = sample_scheme
        if proj_method != 'favorp' or cos_weighting:
            raise NotImplementedError(
                "only proj_method='favorp' without cos_weighting is built for MI355X (SURVEY.md 2.1 #7)")
        self.use_random_proj = True
        mat = create_proj_matrix(self.num_heads, approx_attn_dim, self.head_dim, ortho=True)
        if sample_scheme == 'default':
            self.register_buffer('eval_proj', mat)
        elif sample_scheme == 'fixed':
            self.register_buffer('random_proj', mat)
        elif sample_scheme == 'learnable':
            # the kernels return no gradient for the feature matrix: refuse rather than silently freeze it
            raise NotImplementedError("sample_scheme='learnable' (gradient with respect to the random "
                                      "features) is not built for MI355X")
        else:
            raise NotImplementedError('other sample schemes are not implemented yet.')
        self.apply(self._init_weights)

    def get_proj_matrix(self, device=None, dtype=None):
        if self.sample_scheme == 'default':
            if self.training:
                return create_proj_matrix(self.num_heads, self.approx_attn_dim, self.head_dim,
                                          ortho=False, device=device, dtype=dtype)
            return self.eval_proj
        return self.random_proj

    def project_qkv(self, x):
        """fp32 activations outside autocast stay fp32: the Performer core computes in exact fp32 arithmetic
        (_ops.PerformerF32Fn), as the reference does there (abstract_attention.py:120-133)."""
        # (only for the Performer core itself: a subclass with another `_attend` -- ScatterBrain inherits this method through the
        #  MRO and feeds the window kernels -- takes the 16-bit activations those kernels want: ADVICE r04)
        if (type(self)._attend is KernelizedAttention._attend
                and x.dtype == torch.float32 and not torch.is_autocast_enabled() and x.is_cuda and not _ops.PERFORMER_16BIT
                and self.head_dim == 64 and self.approx_attn_dim <= 96 and self.approx_attn_dim % 16 == 0):
            B, N, C = x.shape
            return _ops.linear(x, self.qkv).reshape(B, N, 3, self.num_heads, C // self.num_heads)
        return super().project_qkv(x)

    project_qkv._ea_builtin = True

    def _core_spec(self, B, N, seq_shape, key_padding_mask, device):
        """The single-node path (_ops.CoreModuleFn, round 4) for the exact-fp32 core; everything else keeps the three nodes."""
        if (type(self)._attend is not KernelizedAttention._attend or _ops.PERFORMER_16BIT or self.head_dim != 64
                or self.approx_attn_dim > 96 or self.approx_attn_dim % 16 != 0):
            return None, ()
        proj = self.get_proj_matrix(device=device, dtype=torch.float32)
        return _ops.PerformerCore(_ops._mask_u8(key_padding_mask, B, N, device), proj), ()

    def _attend(self, qkv5, key_padding_mask, seq_shape):
        B, N = qkv5.shape[:2]
        proj = self.get_proj_matrix(device=qkv5.device, dtype=torch.float32)
        mask = _ops._mask_u8(key_padding_mask, B, N, qkv5.device)
        return _ops.performer_attention(qkv5, mask, proj)

    @staticmethod
    def add_attn_specific_args(parent_parser, struct_name="attn_args", prefix=""):
        parent_parser = MultiheadAttention.add_attn_specific_args(parent_parser, struct_name=struct_name, prefix=prefix)
        group = parent_parser.add_argument_group("Attention")
        fp = prefix + "-" if len(prefix) > 1 else ""
        kw = dict(struct_name=struct_name, prefix=prefix)
        add_nested_argument(group, "--%sapprox-attn-dim" % fp, default=64, type=int,
                            help='number of random features', **kw)
        add_nested_argument(group, "--%sproj-method" % fp, default='favorp', type=str,
                            help='which random feature map is used', **kw)
        add_nested_argument(group, "--%scos-weighting" % fp, action='store_true', default=False, help='', **kw)
        add_nested_argument(group, "--%ssample-scheme" % fp, default='default', type=str, **kw)
        return parent_parser
