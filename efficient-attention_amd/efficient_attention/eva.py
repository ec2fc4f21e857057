"""EVA: attention with control variates (ICLR'23), MI355X build.

Mirrors efficient_attention/eva.py:15-243: `T5RelativePositionBias`, the constructor kwargs
(`adaptive_proj` in {default, no-ln, none}, `num_landmarks`, `use_t5_rpe` + LocalAttention's),
the `adaptive_mu_q/k` parameter layout, the 1-D padding rule of `_process_input`, training-time
sampling `omega = mu + randn_like(mu)` and the argparse flags.  The chunk means, beta, and
the window attention with control-variate columns run in libea_hip.so (_ops.EvaAttnFn).
"""
import math
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import add_nested_argument
from . import _ops
from . import _f32
from .abstract_attention import MultiheadAttention
from .local_attention import LocalAttention


class T5RelativePositionBias(nn.Module):
    """Bucketed relative-position bias (reference :15-65); forward(x) -> [1,h,1,i,j] * scale."""

    def __init__(self, scale, num_heads, causal=False, num_buckets=32, max_distance=128):
        super().__init__()
        self.scale = scale
        self.causal = causal
        self.num_buckets = num_buckets
        self.max_distance = max_distance
        self.relative_attention_bias = nn.Embedding(num_buckets, num_heads)

    @staticmethod
    def _relative_position_bucket(relative_position, causal=True, num_buckets=32, max_distance=128):
        n = -relative_position
        bucket = torch.zeros_like(n)
        if causal:
            n = n.clamp(min=0)
        else:
            num_buckets //= 2
            bucket = bucket + (n < 0).long() * num_buckets
            n = n.abs()
        max_exact = num_buckets // 2
        far = max_exact + (torch.log(n.clamp(min=1).float() / max_exact)
                           / math.log(max_distance / max_exact) * (num_buckets - max_exact)).long()
        far = far.clamp(max=num_buckets - 1)
        return bucket + torch.where(n < max_exact, n, far)

    def bucket_table(self, i, j, device):
        """[i, j] bucket ids.  The float log + truncation sits on exact bucket boundaries for
        some distances, so the table is always evaluated on the CPU (as the reference's CPU path
        does) and cached, never with the device's log implementation."""
        key = (i, j)
        cache = self.__dict__.setdefault("_bucket_cache", {})
        if key not in cache:
            rel = torch.arange(j).view(1, j) - torch.arange(i).view(i, 1)
            cache[key] = self._relative_position_bucket(
                rel, causal=self.causal, num_buckets=self.num_buckets, max_distance=self.max_distance)
        dkey = (i, j, str(device))                      # device copy cached too: an H2D copy from pageable
        if dkey not in cache:                           # memory per forward would also break hipGraph capture
            cache[dkey] = cache[key].to(device)
        return cache[dkey]

    def _bucket_onehot(self, i, j, device):
        """[i*j, num_buckets] fp32 one-hot rows of the bucket table (cached per device)."""
        cache = self.__dict__.setdefault("_bucket_cache", {})
        key = (i, j, str(device), "onehot")
        if key not in cache:
            cache[key] = F.one_hot(self.bucket_table(i, j, device).reshape(-1), self.num_buckets).float()
        return cache[key]

    def dense(self, i, j, device):
        """[h, i, j] bias, already multiplied by scale.  The table lookup is a one-hot GEMM (exact in
        fp32): its backward is the transposed GEMM -- deterministic and safe to capture in a
        hipGraph, which the sort-based embedding backward torch picks above ~3k indices is not."""
        if device.type != "cuda":
            return self.relative_attention_bias(self.bucket_table(i, j, device)).permute(2, 0, 1) * self.scale
        with torch.autocast(device_type="cuda", enabled=False):
            vals = self._bucket_onehot(i, j, device) @ self.relative_attention_bias.weight.float()
        return vals.view(i, j, -1).permute(2, 0, 1) * self.scale

    def table_spec(self, i, j):
        """_ops.TableBias of the [i, j] bias: bias[h, i, j] = scale * weight[bucket(i, j), h] (one launch each way, round 6)."""
        cache = self.__dict__.setdefault("_bucket_cache", {})
        key = (i, j, "spec")
        if key not in cache:
            cache[key] = _ops.TableBias(self.bucket_table(i, j, torch.device("cpu")), self.num_buckets, i, j, self.scale)
        return cache[key]

    def forward(self, x):
        i, j = x.shape[-2:]
        return self.dense(i, j, x.device).unsqueeze(0).unsqueeze(2)


class EVA(LocalAttention):
    def __init__(self, adaptive_proj='default', num_landmarks=49, use_t5_rpe=False, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.adaptive_proj = adaptive_proj
        d = self.head_dim

        def mu_net(with_ln):
            layers = [nn.Linear(d, d)] + ([nn.LayerNorm(d)] if with_ln else [])
            return nn.Sequential(*layers)

        if adaptive_proj == 'default':
            self.adaptive_mu_q, self.adaptive_mu_k = mu_net(True), mu_net(True)
        elif adaptive_proj == 'no-ln':
            self.adaptive_mu_q, self.adaptive_mu_k = mu_net(False), mu_net(False)
        elif adaptive_proj == 'none':
            self.adaptive_mu_k = mu_net(True)
        self.use_t5_rpe = use_t5_rpe
        self.num_landmarks = num_landmarks
        if self.use_rpe and self.use_t5_rpe:
            raise NotImplementedError("Default RPE and T5-style RPE cannot be true simultaneously.")
        if self.use_rpe:
            warnings.warn("--use-rpe selects the default window relative positional embedding; "
                          "the T5-style encoding (--use-t5-rpe alone) usually works slightly better.")
        if self.use_t5_rpe:
            span = self.window_size + self.ext_size
            self.rel_pos_bias = T5RelativePositionBias(
                self.scale, num_heads=self.num_heads, causal=False,
                num_buckets=max(min(int(span / 2), 64), 16), max_distance=span)
        self.apply(self._init_weights)

    def _mu_params(self):
        if self.adaptive_proj == 'default':
            q, k = self.adaptive_mu_q, self.adaptive_mu_k
            return [q[0].weight, q[0].bias, q[1].weight, q[1].bias,
                    k[0].weight, k[0].bias, k[1].weight, k[1].bias]
        if self.adaptive_proj == 'no-ln':
            q, k = self.adaptive_mu_q, self.adaptive_mu_k
            return [q[0].weight, q[0].bias, k[0].weight, k[0].bias]
        k = self.adaptive_mu_k
        return [k[0].weight, k[0].bias, k[1].weight, k[1].bias]

    def _mu_f32(self, q_mean, k_mean):
        """(rf_k_bar, mu) from the chunk means [B,h,L,d] (eva.py:178-185), the module's own Linear / LayerNorm layers in fp32."""
        if self.adaptive_proj in ('default', 'no-ln'):
            rq, rk = self.adaptive_mu_q(q_mean), self.adaptive_mu_k(k_mean)
            return rk, 0.5 * (rq + rk)
        rk = self.adaptive_mu_k(k_mean)
        return rk, torch.zeros_like(rk)

    def _process_input(self, x, key_padding_mask):
        """2-D: validate the grid; 1-D: pad x to a multiple of the window and build/extend the
        padding mask (reference :119-136)."""
        B, *seq_shape, C = x.shape
        w = self.window_size
        if self.attn_2d:
            assert len(seq_shape) == 2
            if w > 0:
                assert seq_shape[0] % w == 0 and seq_shape[1] % w == 0
            return x, key_padding_mask, seq_shape
        n = seq_shape[0]
        n_pad = int(math.ceil(n / w) * w) if w > 0 else n
        if key_padding_mask is None and n_pad == n and _ops.EVA_1D_NO_MASK:
            # nothing padded and nobody masked: no mask at all (round 6) -- an all-false one cost two framework launches per step
            # and kept the window kernels off their static-key-validity instantiations
            return x, None, [n_pad]
        mask = torch.zeros(B, n_pad, dtype=torch.bool, device=x.device)
        if key_padding_mask is not None:
            mask[:, :n] = key_padding_mask.to(torch.bool)
        if n_pad != n:
            x = F.pad(x, (0, 0, 0, n_pad - n))
            mask[:, n:] = True
        return x, mask, [n_pad]

    def forward(self, x, key_padding_mask=None):
        B, *seq_shape, C = x.shape
        orig_n = int(math.prod(seq_shape))
        x, key_padding_mask, seq_shape = self._process_input(x, key_padding_mask)
        N = int(math.prod(seq_shape))
        w, e, h, d = self.window_size, self.ext_size, self.num_heads, self.head_dim
        pooled = None
        # the common training case as ONE autograd node (projections + core, round 4): decided before anything is launched
        if (torch.is_autocast_enabled() and x.is_cuda and (self.proj_drop.p == 0.0 or not self.training)
                and w > 0 and self.num_landmarks > 0 and getattr(type(self).project_qkv, "_ea_builtin", False)
                and type(self).merge_and_project is MultiheadAttention.merge_and_project):
            r0 = int(math.sqrt(N // self.num_landmarks)) if self.attn_2d else int(N // self.num_landmarks)
            ok_geo = r0 > 0 and (e > 0 or (all(s_ % r0 == 0 for s_ in seq_shape) if self.attn_2d else N % r0 == 0))
            L0 = ((seq_shape[0] // r0) * (seq_shape[1] // r0) if self.attn_2d else N // r0) if r0 > 0 else 0
            cdt = torch.get_autocast_dtype("cuda")
            if ok_geo and _ops.eva_module_fn_supported(x, self.qkv, self.proj, cdt, self.adaptive_proj, L0, d):
                Wq_, Wk_ = (w * w, (w + 2 * e) ** 2) if self.attn_2d else (w, w + 2 * e)
                tb = None
                if self.use_t5_rpe and _ops.USE_TABLE_BIAS:
                    bias, tb = self.rel_pos_bias.relative_attention_bias.weight, self.rel_pos_bias.table_spec(Wq_, Wk_)
                elif self.use_t5_rpe:
                    bias = self.rel_pos_bias.dense(Wq_, Wk_, x.device)
                else:
                    bias, tb = self._table_spec()
                    if tb is None:
                        bias = self._table_bias()
                noise = None
                if self.training:
                    noise = torch.randn_like(torch.empty(B, h, L0, d, device=x.device, dtype=torch.float32))
                mask = _ops._mask_u8(key_padding_mask, B, N, x.device)
                cfg = (self.attn_2d, tuple(seq_shape), w, e, r0, L0, self.adaptive_proj) + ((tb,) if tb is not None else ())
                y = _ops.EvaModuleFn.apply(x, self.qkv.weight, self.qkv.bias, self.proj.weight, self.proj.bias, bias, mask, noise,
                                           cfg, cdt, h, *self._mu_params())
                y = self.proj_drop(y)
                if not self.attn_2d:
                    y = y[..., :orig_n, :]
                return y
        if self.attn_2d:
            r = int(math.sqrt(N // self.num_landmarks))
            L = (seq_shape[0] // r) * (seq_shape[1] // r)
            if e == 0:
                assert seq_shape[0] % r == 0 and seq_shape[1] % r == 0
            Wq, Wk = w * w, (w + 2 * e) ** 2
            grid = (B, seq_shape[0], seq_shape[1], r)
            if e == 0 and key_padding_mask is None and L <= 64 and _ops.linear_pool_usable(x, self.qkv, grid, h):
                # the chunk means of q, k (eva.py:178-181 on 2-D chunks without extension) leave the projection kernel
                # with qkv: no second pass over q, k (round 4)
                y, pq, pk = _ops.LinearPoolFn.apply(x.reshape(B, N, C), self.qkv.weight, self.qkv.bias,
                                                    torch.get_autocast_dtype("cuda"), grid)
                qkv5 = y.reshape(B, N, 3, h, d)
                pooled = ("pooled", pq, pk)
        else:
            r = int(N // self.num_landmarks)
            L = N // r
            if e == 0:
                assert N % r == 0
            Wq, Wk = w, w + 2 * e

        if pooled is None:
            qkv5 = self.project_qkv(x.reshape(B, N, C))

        if self.use_t5_rpe:
            bias = self.rel_pos_bias.dense(Wq, Wk, x.device)
        else:
            bias = self._table_bias()
        noise = None
        if self.training:
            noise = torch.randn_like(torch.empty(B, h, L, d, device=x.device, dtype=torch.float32))
        mask = _ops._mask_u8(key_padding_mask, B, N, x.device)
        cfg = (self.attn_2d, tuple(seq_shape), w, e, r, L, self.adaptive_proj)
        if pooled is not None:
            cfg = cfg + (0, 0.5, None, 1.0, pooled)
        if qkv5.dtype == torch.float32:
            # fp32 outside autocast (round 5): the core on the fp32-faithful kernels, the mu networks as the module's own layers
            out = _f32.eva_core(qkv5, bias, noise, mask, self.attn_2d, tuple(seq_shape), w, e, r, self._mu_f32)
        elif L > 64 and _f32.ENABLED:
            # more landmarks than the 16-bit window kernels hold (64): the generic fp32 kernels on the 16-bit activations
            # (exact in fp32) -- the reference takes any --num-landmarks (eva.py:155-164)
            with torch.autocast(device_type="cuda", enabled=False):
                out = _f32.eva_core(qkv5.float(), None if bias is None else bias.float(), noise, mask, self.attn_2d,
                                    tuple(seq_shape), w, e, r, self._mu_f32).to(qkv5.dtype)
        else:
            out = _ops.EvaAttnFn.apply(qkv5, bias, noise, mask, cfg, *self._mu_params())
        y = self.merge_and_project(out, B, seq_shape, C, x.dtype)
        if not self.attn_2d:
            y = y[..., :orig_n, :]
        return y

    @staticmethod
    def add_attn_specific_args(parent_parser, struct_name="attn_args", prefix=""):
        parent_parser = LocalAttention.add_attn_specific_args(parent_parser, struct_name=struct_name, prefix=prefix)
        group = parent_parser.add_argument_group("attention")
        fp = prefix + "-" if len(prefix) > 1 else ""
        kw = dict(struct_name=struct_name, prefix=prefix)
        add_nested_argument(group, "--%sadaptive-proj" % fp, default='default', type=str, **kw)
        add_nested_argument(group, "--%snum-landmarks" % fp, default=49, type=int, **kw)
        add_nested_argument(group, "--%suse-t5-rpe" % fp, action="store_true", default=False, **kw)
        return parent_parser
