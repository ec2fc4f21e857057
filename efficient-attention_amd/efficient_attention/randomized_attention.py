"""Randomized attention (RA, arXiv 2204.04667), MI355X build.

Mirrors efficient_attention/randomized_attention.py:10-63 of the reference: `num_samples` kwarg /
flag, same parameters as the softmax baseline, `forward(x, key_padding_mask=None)` -- and, like the
reference's `_apply_attention`, the padding mask is not used by either softmax.  Per query i:

    num_samples ==  0 : mu_i = q_i + mean_j k_j
    num_samples == -1 : mu_i = q_i + sum_j softmax_j(s q_i.k_j) k_j
    otherwise         : mu_i = q_i + k_J,  J ~ softmax_j(s q_i.k_j)        (one draw, no gradient)
    w_i = mu_i (+ N(0, I) in training);   out_i = sum_j softmax_j(s w_i.k_j - s |k_j|^2 / 2) v_j

Both softmaxes run in the streaming HIP kernels of the softmax baseline (`_ops.SoftmaxQKVFn`: the
first with the keys as values, the second with the per-key norm term and its gradient); the draw is a
Gumbel-max pass over the keys (`_ops.softmax_sample`) instead of `torch.multinomial` on a
materialised [N, N] matrix.  There is no CPU fallback.
"""
import torch

from . import add_nested_argument
from . import _ops
from . import _f32
from .abstract_attention import MultiheadAttention


class RandomizedAttention(MultiheadAttention):
    _graph_core_params = ()             # `_attend` reads no parameter of its own: the single-node path applies (round 6)

    def __init__(self, num_samples=1, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.num_samples = num_samples
        self._sample_index_fn = None                   # tests: injected draws
        self.apply(self._init_weights)

    def _attend_f32(self, qkv5):
        """fp32 activations outside autocast (round 5): both softmax passes on the fp32 gathered-attention kernels
        (randomized_attention.py:21-52 in the precision the reference computes it)."""
        q, k, v = _f32._qkv(qkv5)
        B, h, N, d = q.shape
        all_n = _f32._cached(("all", N, str(q.device)), lambda: torch.arange(N, device=q.device, dtype=torch.int32).view(1, N))
        spec = dict(idx_q=all_n, idx_k=all_n, scale=self.scale)
        if self.num_samples == 0:
            mu = q + k.mean(dim=-2, keepdim=True)
        elif self.num_samples == -1:
            mu = q + _f32.GatherAttnFn.apply(q, k, k, None, None, None, spec)[0]
        else:
            with torch.no_grad():
                if self._sample_index_fn is not None:
                    index = self._sample_index_fn((B, h, N)).to(device=q.device, dtype=torch.int64)
                else:
                    # the draw itself comes from the 16-bit Gumbel-max pass (a sample, not a value: its logits carry bf16 rounding)
                    q16, k16, _ = _ops._qkv_views(qkv5.detach().to(torch.bfloat16))
                    index = _ops.softmax_sample(q16, k16)
            mu = q + torch.gather(k, 2, index.unsqueeze(-1).expand(B, h, N, d))
        w = mu + torch.randn_like(mu) if self.training else mu
        out, _ = _f32.GatherAttnFn.apply(w.contiguous(), k, v, None, None, None, dict(spec, knorm=1))
        return out.permute(0, 2, 1, 3)

    def _attend(self, qkv5, key_padding_mask, seq_shape):
        if qkv5.dtype == torch.float32:
            return self._attend_f32(qkv5)
        q, k, v = _ops._qkv_views(qkv5)                 # [B,h,N,d] views of the projection output
        B, h, N, d = q.shape
        if self.num_samples == 0:
            mu = q.float() + k.float().mean(dim=-2, keepdim=True)
        elif self.num_samples == -1:
            mu = q.float() + _ops.SoftmaxQKVFn.apply(q, k, k, 0).permute(0, 2, 1, 3).float()
        else:
            with torch.no_grad():
                if self._sample_index_fn is not None:
                    index = self._sample_index_fn((B, h, N)).to(device=q.device, dtype=torch.int64)
                else:
                    index = _ops.softmax_sample(q, k)
            mu = q.float() + torch.gather(k, 2, index.unsqueeze(-1).expand(B, h, N, d)).float()
        w = mu + torch.randn_like(mu) if self.training else mu
        return _ops.SoftmaxQKVFn.apply(w.to(qkv5.dtype), k, v, 1)

    @staticmethod
    def add_attn_specific_args(parent_parser, struct_name="attn_args", prefix=""):
        parent_parser = MultiheadAttention.add_attn_specific_args(parent_parser, struct_name=struct_name, prefix=prefix)
        group = parent_parser.add_argument_group("Attention")
        fp = prefix + "-" if len(prefix) > 1 else ""
        add_nested_argument(group, "--%snum-samples" % fp, struct_name=struct_name, prefix=prefix, default=1,
                            type=int, help="number of random features")
        return parent_parser
