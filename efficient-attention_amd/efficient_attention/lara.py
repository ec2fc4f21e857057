"""LARA: linear randomized attention (ICML'22), MI355X build.

Mirrors efficient_attention/lara.py:14-267: constructor kwargs, the `q_bar_gen/k_bar_gen`
Sequential layouts (so state_dict keys are `q_bar_gen.{2,3}.*` for pooled and `{0,1}.*` for
adaptive-1d proposals), the 2-D / 1-D dispatch on the rank of `x`, the sampling modes and the
argparse flags.  The O(N L d) estimator -- random-feature projections of q and k, the
per-landmark softmax statistics over the sequence, the importance weights and the final
combine -- runs in libea_hip.so (_ops.LaraAttnFn).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import add_nested_argument
from . import _ops
from . import _f32
from .abstract_attention import MultiheadAttention
from .attn_utils import FlattenTranspose


class LinearRA(_ops.DerivedCacheOwner, MultiheadAttention):
    def __init__(self, num_landmarks=49, kernel_size=None, proposal_gen='pool',
                 use_antithetics=False, use_multisample=False, pool_module_type='light',
                 mis_type='mis-opt', alpha_coeff=1.0, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.num_landmarks = num_landmarks
        self.proposal_gen = proposal_gen
        self.use_antithetics = use_antithetics
        self.use_multisample = use_multisample
        self.pool_module_type = pool_module_type
        self.mis_type = mis_type
        self.alpha_coeff = alpha_coeff
        if pool_module_type == 'dense':
            ch = self.dim
        elif pool_module_type == 'light':
            ch = self.head_dim
        side = int(math.sqrt(num_landmarks))

        def gen():
            if proposal_gen.startswith('pool'):
                return nn.Sequential(nn.AdaptiveAvgPool2d(side), FlattenTranspose(),
                                     nn.Linear(ch, ch), nn.LayerNorm(ch))
            if proposal_gen.startswith('no-param-pool'):
                return nn.Sequential(nn.AdaptiveAvgPool2d(side), FlattenTranspose())
            if proposal_gen.startswith('adaptive-1d'):
                return nn.Sequential(nn.Linear(ch, ch), nn.LayerNorm(ch))
            raise NotImplementedError

        self.q_bar_gen = gen()
        self.k_bar_gen = gen()
        self.apply(self._init_weights)

    # ---- landmark proposals (tiny [B,h,L,d] tensors; pooling reads q,k once) -------------
    def _proposal_gen_2d(self, qkv5, H, W, slot=None, mix=True, pooled=None):
        """Adaptive 2-D average pool of q,k -> [Linear+LN] -> optional softmax mixing of k_bar
        (reference :129-175).  Returns q_bar, k_bar [B,h,L,d] fp32 and the '-vmixed' column bias of the
        mixing logits (or None); with mix=False the mixing itself is left to the landmark kernel."""
        B, N, _, h, d = qkv5.shape
        side = int(math.sqrt(self.num_landmarks))
        gen = self.proposal_gen
        # pooled: (pq, pk, pv | None) [B,h,L,d] computed by the caller (the fp32 path pools with nn.AdaptiveAvgPool2d itself)
        pq, pk, pv = pooled if pooled is not None else _ops.pool2d_qkv(qkv5, H, W, side, slot, need_v=gen.endswith('-vmixed'))
        if gen.startswith('pool'):
            if self.pool_module_type == 'dense':
                def dense(p, net):
                    z = p.permute(0, 2, 1, 3).reshape(B, side * side, h * d)
                    # model-wide Linear through the projection kernels (ea_linear / ea_wgrad), its LayerNorm over the
                    # B * L rows of `dim` channels through ea_layernorm_fwd / _bwd (round 4)
                    z = _ops.layer_norm(_ops.linear(z.contiguous(), net[2]), net[3])
                    return z.reshape(B, side * side, h, d).permute(0, 2, 1, 3)
                q_bar, k_bar = dense(pq, self.q_bar_gen), dense(pk, self.k_bar_gen)
            else:
                q_bar = self.q_bar_gen[3](self.q_bar_gen[2](pq))
                k_bar = self.k_bar_gen[3](self.k_bar_gen[2](pk))
        else:
            q_bar, k_bar = pq, pk
        q_bar, k_bar = q_bar.float(), k_bar.float()
        colbias = torch.log(pv.float().norm(dim=-1) + 1e-4) if gen.endswith('-vmixed') else None      # [B,h,L]
        if gen.endswith('mixed') and mix:
            logits = self.scale * torch.einsum('bhpd,bhcd->bhpc', k_bar, k_bar)
            if colbias is not None:
                logits = logits + colbias.unsqueeze(-2)
            k_bar = torch.einsum('bhpc,bhcd->bhpd', torch.softmax(logits, dim=-1), k_bar)
        return q_bar, k_bar, colbias

    def _proposal_gen_1d(self, qkv5, key_padding_mask):
        """Segment means of (optionally Linear+LN'd) q,k with the even / uneven split rule
        (reference :84-127).  Returns q_bar, k_bar [B,h,L,d] fp32 and the (mask-zeroed) qkv."""
        B, N, _, h, d = qkv5.shape
        L = self.num_landmarks
        if key_padding_mask is not None:
            keep = (~key_padding_mask.to(torch.bool)).to(qkv5.dtype).view(B, N, 1, 1, 1)
            qkv5 = qkv5 * keep
        q, k, _ = _ops._qkv_views(qkv5)
        if self.proposal_gen.startswith('adaptive-1d'):
            q2, k2 = self.q_bar_gen(q), self.k_bar_gen(k)
        else:
            q2, k2 = q, k
        q2, k2 = q2.float(), k2.float()
        if N <= L:
            return q2, k2, qkv5
        segs = N // L
        if N % L == 0:
            q_bar = q2.reshape(B, h, L, segs, d).mean(-2)
            k_bar = k2.reshape(B, h, L, segs, d).mean(-2)
        else:
            short = (segs + 1) * L - N

            def seg_mean(t):
                head = t[:, :, :short * segs].reshape(B, h, short, segs, d).mean(-2)
                tail = t[:, :, short * segs:].reshape(B, h, L - short, segs + 1, d).mean(-2)
                return torch.cat([head, tail], dim=-2)
            q_bar, k_bar = seg_mean(q2), seg_mean(k2)
        return q_bar, k_bar, qkv5

    # ---- 'adaptive-1d' with the generators' Linear folded into the qkv projection --------------
    def _project_qkv_folded(self, x):
        """x [B,N,C] -> [B,N,5,h,d]: q, k, v and, in slots 3 / 4, Linear_q(q), Linear_k(k) without
        their biases (W' = W_gen W_q per head: the composition of the two projections is one more
        group of output columns of the same GEMM instead of a K = d GEMM over B*h*N strided rows)."""
        B, N, C = x.shape
        h, d = self.num_heads, self.head_dim
        if (_ops.USE_FOLD_KERNELS and x.is_cuda and torch.is_autocast_enabled() and d == 64
                and torch.get_autocast_dtype("cuda") in (torch.bfloat16, torch.float16)
                and not torch.compiler.is_compiling() and torch._C._len_torch_dispatch_stack() == 0):
            # extended weight built (and differentiated) by one HIP launch each way: ea_lara_fold_fwd / _bwd
            lq, lk = self.q_bar_gen[0], self.k_bar_gen[0]
            qkv, bias_q, bias_k = _ops.FoldedQkvFn.apply(x, self.qkv.weight, self.qkv.bias, lq.weight, lq.bias, lk.weight, lk.bias,
                                                         torch.get_autocast_dtype("cuda"), h)
            self._fold_bias = (bias_q, bias_k)
            return qkv.reshape(B, N, 5, h, d)
        self._fold_bias = None

        def build():
            with torch.autocast(device_type="cuda", enabled=False):
                W = self.qkv.weight.float()
                Wq, Wk = W[:C].view(h, d, C), W[C:2 * C].view(h, d, C)
                # W' = W_gen W_head for every head as ONE [d, d] x [d, h*C] product per side (the broadcast form
                # [d,d] x [h,d,C] runs as h tiny batched GEMMs: 40 us each way at h = 8, C = 512)
                def fold(gen_w, Wh):
                    flat = Wh.permute(1, 0, 2).reshape(d, h * C)
                    return (gen_w.float() @ flat).view(d, h, C).permute(1, 0, 2).reshape(C, C)
                W2q = fold(self.q_bar_gen[0].weight, Wq)
                W2k = fold(self.k_bar_gen[0].weight, Wk)
                w_ext = torch.cat([W, W2q, W2k], 0)
                b_ext = None
                if self.qkv.bias is not None:
                    b_ext = torch.cat([self.qkv.bias.float(), self.qkv.bias.new_zeros(2 * C, dtype=torch.float32)])
            return w_ext, b_ext
        if not hasattr(self, "_folded_cache"):
            self._folded_cache = _ops.DerivedCache()
        w_ext, b_ext = self._folded_cache.get(
            self, [self.qkv.weight, self.qkv.bias, self.q_bar_gen[0].weight, self.k_bar_gen[0].weight], build)
        qkv = _ops.linear_wb(x, w_ext, b_ext)
        qkv = _ops.to_io_dtype(qkv)
        return qkv.reshape(B, N, 5, h, d)

    def _proposal_gen_1d_folded(self, qkvE, key_padding_mask, mask_u8, slot):
        """Segment means of LayerNorm(Linear(q)), LayerNorm(Linear(k)) (reference :84-127) from the
        folded projection.  Returns q_bar, k_bar [B,h,L,d] fp32 and the (mask-zeroed) qkvE."""
        B, N, _, h, d = qkvE.shape
        if key_padding_mask is not None:
            keep = (~key_padding_mask.to(torch.bool)).to(qkvE.dtype).view(B, N, 1, 1, 1)
            qkvE = qkvE * keep
        lq, nq, lk, nk = self.q_bar_gen[0], self.q_bar_gen[1], self.k_bar_gen[0], self.k_bar_gen[1]
        fold_bias, self._fold_bias = getattr(self, "_fold_bias", None), None
        if fold_bias is not None:
            bias_q, bias_k = fold_bias
            pq, pk = _ops.SegmentLnMeanFn.apply(qkvE, mask_u8, self.num_landmarks, slot, bias_q, bias_k,
                                                lq.bias, lk.bias, nq.weight, nq.bias, nk.weight, nk.bias)
            return pq, pk, qkvE
        with torch.autocast(device_type="cuda", enabled=False):
            C = h * d
            if self.qkv.bias is not None:
                bq, bk = self.qkv.bias[:C].float().view(h, d), self.qkv.bias[C:2 * C].float().view(h, d)
                bias_q = bq @ lq.weight.float().t() + lq.bias.float()
                bias_k = bk @ lk.weight.float().t() + lk.bias.float()
            else:
                bias_q = lq.bias.float().expand(h, d)
                bias_k = lk.bias.float().expand(h, d)
        pq, pk = _ops.SegmentLnMeanFn.apply(qkvE, mask_u8, self.num_landmarks, slot, bias_q, bias_k,
                                            lq.bias, lk.bias, nq.weight, nq.bias, nk.weight, nk.bias)
        return pq, pk, qkvE

    def _forward_f32(self, x, key_padding_mask, qkv5=None):
        """fp32 activations outside autocast: fp32 Linear layers, the proposals as the reference forms them (adaptive pooling /
        segment means, the generators' own Linear + LayerNorm, softmax mixing: tiny [B,h,L,d] tensors), the estimator on the
        fp32 gathered-attention kernels (_f32.lara_core) -- lara.py:129-251 in the precision the reference computes it."""
        B, *seq_shape, C = x.shape
        N = int(math.prod(seq_shape))
        h, d = self.num_heads, self.head_dim
        if qkv5 is None or qkv5.shape[2] != 3:
            qkv5 = self.project_qkv(x.reshape(B, N, C))
        io = qkv5.dtype
        with torch.autocast(device_type="cuda", enabled=False):
            out = self._core_f32(qkv5.float(), key_padding_mask, seq_shape).to(io)
        return self.merge_and_project(out, B, seq_shape, C, x.dtype)

    def _core_f32(self, qkv5, key_padding_mask, seq_shape):
        B, N, _, h, d = qkv5.shape
        mask = _ops._mask_u8(key_padding_mask, B, N, qkv5.device)
        mode = 0
        if self.training:
            mode = 2 if self.use_multisample else (1 if self.use_antithetics else 0)
        if len(seq_shape) == 2:
            H, W = seq_shape
            side = int(math.sqrt(self.num_landmarks))
            need_v = self.proposal_gen.endswith('-vmixed')

            def pool(t):                                  # t [B,N,h,d] -> [B,h,side*side,d] (lara.py:43,48,145-151)
                m = t.permute(0, 2, 3, 1).reshape(B * h, d, H, W)
                return F.adaptive_avg_pool2d(m, side).reshape(B, h, d, side * side).transpose(-1, -2)
            pooled = (pool(qkv5[:, :, 0]), pool(qkv5[:, :, 1]), pool(qkv5[:, :, 2]) if need_v else None)
            q_bar, k_bar, _ = self._proposal_gen_2d(qkv5, H, W, None, mix=True, pooled=pooled)
        else:
            q_bar, k_bar, qkv5 = self._proposal_gen_1d(qkv5, key_padding_mask)
        noise = None
        if self.training:
            nl = q_bar.shape[-2]
            if self.use_multisample:
                noise = torch.randn(B, h, nl * 2, d, dtype=torch.float32, device=qkv5.device)
            else:
                noise = torch.randn_like(torch.empty(B, h, nl, d, dtype=torch.float32, device=qkv5.device))
        return _f32.lara_core(qkv5, mask, q_bar, q_bar + k_bar, noise, self.mis_type, float(self.alpha_coeff), mode, float(self.scale))

    def _mlp_params(self):
        q, k = self.q_bar_gen, self.k_bar_gen
        return [q[2].weight, q[2].bias, q[3].weight, q[3].bias, k[2].weight, k[2].bias, k[3].weight, k[3].bias]

    def forward(self, x, key_padding_mask=None):
        B, *seq_shape, C = x.shape
        N = int(math.prod(seq_shape))
        h, d = self.num_heads, self.head_dim
        L = self.num_landmarks
        gen = self.proposal_gen
        # 'adaptive-1d' on the GPU.  Round 4: generator Linear + LayerNorm + segment mean in one HIP pass over the stored
        # q / k rows (_ops.SegLinLnMeanFn; d = 64), the projection stays 3C wide.  Otherwise (rounds 1-3) the per-token Linear
        # of q_bar_gen / k_bar_gen rides along in the qkv GEMM (two more groups of output columns).
        # fp32 activations outside autocast (round 5): fp32 end to end -- _forward_f32
        if self._F32_CORE and _f32.usable(x) and d in (32, 64, 128) and len(seq_shape) in (1, 2):
            return self._forward_f32(x, key_padding_mask)
        ad1d = (len(seq_shape) == 1 and gen.startswith('adaptive-1d') and N > L and x.is_cuda)
        # (ea_lara_seglin_* evaluates nn.LayerNorm with its default eps = 1e-5: another eps keeps the folded path -- ADVICE r04)
        seglin = (ad1d and d == 64 and _ops.USE_SEGLIN and self.q_bar_gen[1].eps == 1e-5 and self.k_bar_gen[1].eps == 1e-5
                  and _ops._DIRECT and not torch.compiler.is_compiling() and torch._C._len_torch_dispatch_stack() == 0)
        fold_1d = ad1d and not seglin and d in (32, 64)
        # the common 2-D training case as ONE autograd node (projections + core): decided before anything is launched
        module_fn = (len(seq_shape) == 2 and not fold_1d and torch.is_autocast_enabled()
                     and getattr(type(self).project_qkv, "_ea_builtin", False)
                     and type(self).merge_and_project is MultiheadAttention.merge_and_project
                     and (self.proj_drop.p == 0.0 or not self.training)
                     and _ops.lara_module_fn_supported(x, self.qkv, self.proj, torch.get_autocast_dtype("cuda")))
        dup = self.training and (self.use_multisample or self.use_antithetics)
        # more samples than the 16-bit estimator kernels hold (128): decided BEFORE a projection is chosen, so that the plain
        # 3-slot projection runs once (the folded 5-slot one would be thrown away and redone)
        side = int(math.sqrt(L))
        n_lm = side * side if len(seq_shape) == 2 else min(L, N)
        over128 = n_lm * (2 if dup else 1) > 128
        if over128 and not (_f32.ENABLED and x.is_cuda and d in (32, 64, 128)):
            raise NotImplementedError(
                "LinearRA: %d samples exceed the 128 the 16-bit estimator kernels hold and the generic fp32 kernels are "
                "%s" % (n_lm * (2 if dup else 1), "switched off (EA_F32_CORES=0)" if not _f32.ENABLED else
                        "not available for head_dim %d / this device" % d))
        qkv5 = None
        if over128:
            fold_1d = module_fn = False
        # 'adaptive-1d' on the segment kernels as ONE autograd node as well (round 6, _ops.GraphCore inside CoreModuleFn): the
        # projections of the three-node path ran two weight-gradient launches and two partial sums of their own
        fits_fused = n_lm * (2 if dup else 1) <= 64
        if (seglin and not over128 and fits_fused and _ops.USE_LARA_1D_MODULE_FN and torch.is_autocast_enabled()
                and getattr(type(self).project_qkv, "_ea_builtin", False)
                and type(self).merge_and_project is MultiheadAttention.merge_and_project
                and (self.proj_drop.p == 0.0 or not self.training)
                and _ops.core_module_fn_supported(x, self.qkv, self.proj, torch.get_autocast_dtype("cuda"))):
            mode1 = (2 if self.use_multisample else (1 if self.use_antithetics else 0)) if self.training else 0
            return self._forward_1d_module(x, key_padding_mask, B, N, C, mode1)
        if not module_fn:
            qkv5 = self._project_qkv_folded(x.reshape(B, N, C)) if fold_1d else self.project_qkv(x.reshape(B, N, C))
        mode = 0
        if self.training:
            mode = 2 if self.use_multisample else (1 if self.use_antithetics else 0)
        slot = _ops._GradSlot()
        mask = _ops._mask_u8(key_padding_mask, B, N, x.device)

        # ---- landmark proposals.  Fused HIP pipeline whenever the sample count fits (C <= 64) ----
        if over128:
            # the generic fp32 kernels on the 16-bit activations (exact in fp32) -- the reference takes any --num-landmarks
            # (lara.py:188-196)
            return self._forward_f32(x, key_padding_mask, qkv5=qkv5)
        fused_b = n_lm * (2 if dup else 1) <= 64 and d in (32, 64)
        fused_a = (fused_b and len(seq_shape) == 2 and self.pool_module_type == 'light'
                   and not gen.endswith('-vmixed') and (gen.startswith('pool') or gen.startswith('no-param-pool'))
                   and seq_shape[0] % side == 0 and seq_shape[1] % side == 0
                   and seq_shape[0] // side == seq_shape[1] // side)

        def draw_noise(nl):
            if not self.training:
                return None
            if self.use_multisample:
                return torch.randn(B, h, nl * 2, d, dtype=torch.float32, device=x.device)
            return torch.randn_like(torch.empty(B, h, nl, d, dtype=torch.float32, device=x.device))

        if fused_a:
            # pooling + landmark pipeline + estimator as one autograd node (_ops.LaraPooledFn)
            params = self._mlp_params() if gen.startswith('pool') else ()
            noise = draw_noise(n_lm)
            cfg = (seq_shape[0], seq_shape[1], seq_shape[0] // side, bool(params), gen.endswith('mixed'),
                   _ops.MIS[self.mis_type], mode if noise is not None else 0, float(self.alpha_coeff), float(self.scale))
            if module_fn:
                y = _ops.LaraModuleFn.apply(x, self.qkv.weight, self.qkv.bias, self.proj.weight, self.proj.bias, mask, noise, cfg,
                                            torch.get_autocast_dtype("cuda"), h, *params)
                return self.proj_drop(y)
            out = _ops.LaraPooledFn.apply(qkv5, mask, noise, cfg, *params)
            return self.merge_and_project(out, B, seq_shape, C, x.dtype)
        if module_fn:                                    # (decided for a geometry the fused pipeline does not cover after all)
            qkv5 = self.project_qkv(x.reshape(B, N, C))

        mixed_k = colbias = None
        if len(seq_shape) == 2:
            # the softmax mixing of k_bar runs inside the landmark kernel whenever that kernel is used
            pq, pk, colbias = self._proposal_gen_2d(qkv5, seq_shape[0], seq_shape[1], slot, mix=not fused_b)
            mixed_k = fused_b and gen.endswith('mixed')
        elif seglin:
            if key_padding_mask is not None:           # padded tokens: q = k = v = 0 for the generator AND the estimator (:93-96)
                keep = (~key_padding_mask.to(torch.bool)).to(qkv5.dtype).view(B, N, 1, 1, 1)
                qkv5 = qkv5 * keep
            lq, nq, lk, nk = self.q_bar_gen[0], self.q_bar_gen[1], self.k_bar_gen[0], self.k_bar_gen[1]
            pq, pk = _ops.SegLinLnMeanFn.apply(qkv5, L, slot, lq.weight, lq.bias, lk.weight, lk.bias,
                                               nq.weight, nq.bias, nk.weight, nk.bias)
            # the segment backward runs after the estimator's and rewrites the same dq rows: it applies the estimator's last
            # correction too (ea_lara_seglin_bwd_fin) whenever it is going to run, i.e. the rows it reads carry a gradient
            slot.defer_fin = bool(_ops.USE_SEGLIN_FIN and fused_b and pq.requires_grad and torch.is_grad_enabled())
        elif fold_1d:
            pq, pk, qkv5 = self._proposal_gen_1d_folded(qkv5, key_padding_mask, mask, slot)
        elif len(seq_shape) == 1:
            pq, pk, qkv5 = self._proposal_gen_1d(qkv5, key_padding_mask)
        else:
            raise ValueError("LinearRA expects x of rank 3 or 4")
        noise = draw_noise(pq.shape[-2])
        if fused_b:
            omega, qrows, bhv, lp = _ops.lara_landmarks(pq, pk, noise, self.mis_type, mode, self.scale, None,
                                                        bool(mixed_k), colbias if mixed_k else None)
            out = _ops.LaraAttnFn.apply(qkv5, mask, omega, qrows, bhv, lp, _ops.MIS[self.mis_type],
                                        float(self.alpha_coeff), slot)
        else:
            out = _ops.lara_attention(qkv5, mask, pq, pq + pk, noise, self.mis_type, self.alpha_coeff,
                                      mode, self.scale, slot)
        return self.merge_and_project(out, B, seq_shape, C, x.dtype)

    def _forward_1d_module(self, x, key_padding_mask, B, N, C, mode):
        """'adaptive-1d' proposals + estimator between the two projections of ONE autograd node (_ops.CoreModuleFn with a
        _ops.GraphCore): the same kernels in the same order as the three-node path below it in forward()."""
        h, d, L = self.num_heads, self.head_dim, self.num_landmarks
        mask = _ops._mask_u8(key_padding_mask, B, N, x.device)
        noise = None
        if self.training:                                # (the one sampling call of the step, shapes as in forward())
            if self.use_multisample:
                noise = torch.randn(B, h, L * 2, d, dtype=torch.float32, device=x.device)
            else:
                noise = torch.randn_like(torch.empty(B, h, L, d, dtype=torch.float32, device=x.device))
        mis, kappa, scale, mis_type = _ops.MIS[self.mis_type], float(self.alpha_coeff), self.scale, self.mis_type

        def core(qkv5, lqw, lqb, lkw, lkb, nqw, nqb, nkw, nkb):
            slot = _ops._GradSlot()
            if key_padding_mask is not None:           # padded tokens: q = k = v = 0 for the generator AND the estimator (:93-96)
                keep = (~key_padding_mask.to(torch.bool)).to(qkv5.dtype).view(B, N, 1, 1, 1)
                qkv5 = qkv5 * keep
            pq, pk = _ops.SegLinLnMeanFn.apply(qkv5, L, slot, lqw, lqb, lkw, lkb, nqw, nqb, nkw, nkb)
            slot.defer_fin = bool(_ops.USE_SEGLIN_FIN and pq.requires_grad and torch.is_grad_enabled())
            omega, qrows, bhv, lp = _ops.lara_landmarks(pq, pk, noise, mis_type, mode, scale, None, False, None)
            return _ops.LaraAttnFn.apply(qkv5, mask, omega, qrows, bhv, lp, mis, kappa, slot)

        lq, nq, lk, nk = self.q_bar_gen[0], self.q_bar_gen[1], self.k_bar_gen[0], self.k_bar_gen[1]
        params = (lq.weight, lq.bias, lk.weight, lk.bias, nq.weight, nq.bias, nk.weight, nk.bias)
        need_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        y = _ops.CoreModuleFn.apply(x, self.qkv.weight, self.qkv.bias, self.proj.weight, self.proj.bias,
                                    _ops.GraphCore(core, len(params), need_grad), torch.get_autocast_dtype("cuda"), h, *params)
        return self.proj_drop(y)

    @staticmethod
    def add_attn_specific_args(parent_parser, struct_name="attn_args", prefix=""):
        parent_parser = MultiheadAttention.add_attn_specific_args(parent_parser, struct_name=struct_name, prefix=prefix)
        group = parent_parser.add_argument_group("attention")
        fp = prefix + "-" if len(prefix) > 1 else ""
        kw = dict(struct_name=struct_name, prefix=prefix)
        add_nested_argument(group, "--%snum-landmarks" % fp, default=49, type=int, **kw)
        add_nested_argument(group, "--%skernel-size" % fp, default=None, type=int, **kw)
        add_nested_argument(group, "--%spool-module-type" % fp, default='light', type=str, **kw)
        add_nested_argument(group, "--%smis-type" % fp, default='mis-opt', type=str, **kw)
        add_nested_argument(group, "--%sproposal-gen" % fp, default='pool', type=str, **kw)
        add_nested_argument(group, "--%suse-antithetics" % fp, action='store_true', default=False, **kw)
        add_nested_argument(group, "--%suse-multisample" % fp, action='store_true', default=False, **kw)
        add_nested_argument(group, "--%salpha-coeff" % fp, default=1.0, type=float, **kw)
        return parent_parser
