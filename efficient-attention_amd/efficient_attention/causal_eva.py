"""Causal EVA for autoregressive language modelling (fairseq decoder self-attention), MI355X build.

Mirrors efficient_attention/causal_eva.py:301-927 of the reference: the fairseq
MultiheadAttention-style constructor (`embed_dim, num_heads, kdim, vdim, dropout, bias,
self_attention, q_noise, qn_block_size, attn_args`), the parameters `q_proj / k_proj / v_proj /
out_proj`, `adaptive_mu_q/k` and the single-head `rel_pos_bias`, time-first
`forward(query, key, value, key_padding_mask=None, incremental_state=None, ...) -> (out, None)`,
the incremental-state helpers fairseq's decoder calls, and the argparse flags.

The training / evaluation path (reference :666-790: chunk means -> mu -> beta, window attention
whose extension lies on the left only, padded queries masked, causal local mask and per-chunk causal
control-variate mask under one softmax) runs in libea_hip.so through `_ops.EvaAttnFn` with
`ea_geom.causal` set, attention dropout included (keep mask drawn here, applied in the kernels);
there is no CPU fallback.  Quantization noise on the projections (`q_noise > 0`, reference :118-213) is applied
to the parameters in front of the projection kernels exactly as the reference's forward pre-hooks do.  Token-by-token
decoding with an incremental state (reference :542-665): `_decode`.
"""
import math
import uuid

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import add_nested_argument
from . import _ops
from . import _f32
from .eva import T5RelativePositionBias


def _mu_net(d, with_ln):
    return nn.Sequential(*([nn.Linear(d, d)] + ([nn.LayerNorm(d)] if with_ln else [])))


class CausalEVAttention(_ops.DerivedCacheOwner, nn.Module):
    def __init__(self, embed_dim, num_heads, kdim=None, vdim=None, dropout=0.0, bias=True,
                 self_attention=False, q_noise=0.0, qn_block_size=8, attn_args=None):
        super().__init__()
        self._incremental_state_id = str(uuid.uuid4())
        self.embed_dim = embed_dim
        self.kdim = embed_dim if kdim is None else kdim
        self.vdim = embed_dim if vdim is None else vdim
        self.qkv_same_dim = self.kdim == embed_dim and self.vdim == embed_dim
        self.num_heads = num_heads
        self.dropout_module = nn.Dropout(dropout)       # holds p (fairseq reads it); applied inside the kernels
        self.head_dim = embed_dim // num_heads
        assert self.head_dim * num_heads == embed_dim, "embed_dim must be divisible by num_heads"
        self.scaling = self.head_dim ** -0.5
        self.self_attention = self_attention
        assert not self_attention or self.qkv_same_dim, \
            "Self-attention requires query, key and value to be of the same size"
        # quantization noise (reference :118-213, 339-351: fairseq's quant_noise wrapper around the four projections)
        self.q_noise, self.qn_block_size = float(q_noise), int(qn_block_size)
        if self.q_noise > 0:
            for in_features in (embed_dim, self.kdim, self.vdim):
                assert in_features % self.qn_block_size == 0, "Input features must be a multiple of block sizes"
        self.k_proj = nn.Linear(self.kdim, embed_dim, bias=bias)
        self.v_proj = nn.Linear(self.vdim, embed_dim, bias=bias)
        self.q_proj = nn.Linear(embed_dim, embed_dim, bias=bias)
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=bias)

        self.window_size = attn_args.window_size
        self.ext_size = max(1, self.window_size) if attn_args.overlap_window else 0
        self.causal = attn_args.causal
        self.num_chunks = attn_args.num_chunks
        self.chunk_size = attn_args.chunk_size
        if self.chunk_size is not None:
            assert self.window_size >= self.chunk_size and self.window_size % self.chunk_size == 0
            self.num_chunks = None                      # chunk_size overrides the number of landmarks
        self.use_t5_rpe = attn_args.use_t5_rpe if attn_args.window_size > 0 else False
        if self.use_t5_rpe:
            span = self.window_size + self.ext_size
            self.rel_pos_bias = T5RelativePositionBias(
                self.scaling, num_heads=1, causal=self.causal,
                num_buckets=max(min(int(span / 2), 64), 16), max_distance=span)
        else:
            self.rel_pos_bias = None
        self.adaptive_proj = attn_args.adaptive_proj
        if self.adaptive_proj in ("qk", "no-ln"):
            ln = self.adaptive_proj == "qk"
            self.adaptive_mu_q = _mu_net(self.head_dim, ln)
            self.adaptive_mu_k = _mu_net(self.head_dim, ln)
        self.reset_parameters()
        self.onnx_trace = False
        self._keep_mask_fn = None
        self._qnoise_mask_fn = None                     # tests: block-drop decisions of a fixture instead of bernoulli_

    # ---- quantization noise (reference :165-213) -------------------------------------------
    def _quant_noise_(self, lin):
        """What the reference's forward pre-hook does to a projection before every TRAINING-mode call: each run of
        `qn_block_size` consecutive input features of each output row is zeroed with probability q_noise, the rest scaled by
        1 / (1 - q_noise), and the result written through `weight.data` -- the parameter itself changes and receives the
        gradient with respect to the noised values.  Parameter-sized work (one draw + one masked scale per projection) in
        front of the projection kernels, which then read the weight as they always do."""
        p = self.q_noise
        if not (self.training and p > 0):
            return
        w = lin.weight
        out_f, in_f = w.shape
        bs = self.qn_block_size
        n = in_f // bs * out_f
        with torch.no_grad():
            if self._qnoise_mask_fn is not None:
                mask = self._qnoise_mask_fn(n).to(device=w.device, dtype=torch.float32).reshape(-1)
            else:
                mask = torch.zeros(n, device=w.device)
                mask.bernoulli_(p)
            mask = mask.repeat_interleave(bs, -1).view(-1, in_f).to(torch.bool)
            w.data = (1.0 / (1.0 - p)) * w.data.masked_fill(mask, 0)

    # ---- initialisation (reference :397-424) ----------------------------------------------
    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            nn.init.xavier_uniform_(m.weight, gain=1 / math.sqrt(2))
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def reset_parameters(self):
        gain = 1 / math.sqrt(2) if self.qkv_same_dim else 1.0
        for proj in (self.k_proj, self.v_proj, self.q_proj):
            nn.init.xavier_uniform_(proj.weight, gain=gain)
        for name in ("adaptive_mu_q", "adaptive_mu_k"):
            if hasattr(self, name):
                getattr(self, name).apply(self._init_weights)
        nn.init.xavier_uniform_(self.out_proj.weight)
        if self.out_proj.bias is not None:
            nn.init.constant_(self.out_proj.bias, 0.0)

    def prepare_for_onnx_export_(self):
        self.onnx_trace = True

    # ---- forward ---------------------------------------------------------------------------
    def _mu_params(self):
        q, k = self.adaptive_mu_q, self.adaptive_mu_k
        if self.adaptive_proj == "qk":
            return [q[0].weight, q[0].bias, q[1].weight, q[1].bias,
                    k[0].weight, k[0].bias, k[1].weight, k[1].bias]
        return [q[0].weight, q[0].bias, k[0].weight, k[0].bias]

    def _project(self, query, key, value, keep_f32=False):
        """Time-first [N, B, C] inputs -> fused [N, B, 3, h, d] in the kernels' I/O dtype."""
        N, B, C = query.shape
        for lin in (self.q_proj, self.k_proj, self.v_proj):       # the hooks' firing order (reference :511-518)
            self._quant_noise_(lin)
        qkv = None
        if self.self_attention and (self.training or torch.is_grad_enabled()):
            # wide layers whenever the derived-weight cache below cannot be used (training, or autograd on; round 6): the stacked
            # 16-bit operand straight from the three master weights, no fp32 concatenation in between (_ops.StackedLinearFn);
            # None = not applicable
            qkv = _ops.linear_stacked(query, [self.q_proj, self.k_proj, self.v_proj])
        if qkv is not None:
            pass
        elif self.self_attention:
            # one GEMM over the stacked weights instead of three over the same activations
            biases = [self.q_proj.bias, self.k_proj.bias, self.v_proj.bias]

            def build():
                return (torch.cat([self.q_proj.weight, self.k_proj.weight, self.v_proj.weight], 0),
                        None if biases[0] is None else torch.cat(biases, 0))
            if not hasattr(self, "_stacked_cache"):
                self._stacked_cache = _ops.DerivedCache()
            weight, bias = self._stacked_cache.get(
                self, [self.q_proj.weight, self.k_proj.weight, self.v_proj.weight] + biases, build)
            qkv = _ops.linear_wb(query, weight, bias)
        else:
            assert key is not None and value is not None
            assert key.shape[:2] == query.shape[:2] and value.shape[:2] == query.shape[:2], \
                "the windowed path needs keys/values aligned with the queries"
            qkv = torch.stack([_ops.linear(query, self.q_proj), _ops.linear(key, self.k_proj),
                               _ops.linear(value, self.v_proj)], dim=2)
        if not (keep_f32 and _f32.usable(qkv) and self.head_dim in (32, 64, 128)):
            qkv = _ops.to_io_dtype(qkv)
        return qkv.reshape(N, B, 3, self.num_heads, self.head_dim)

    def forward(self, query, key, value, key_padding_mask=None, incremental_state=None,
                need_weights=True, attn_mask=None):
        """query/key/value: Time x Batch x Channel; key_padding_mask: [B, T], 1 = pad.
        Returns (out [T, B, C], None)."""
        if incremental_state is not None:
            return self._decode(query, key_padding_mask, incremental_state)
        if self.adaptive_proj not in ("qk", "no-ln"):
            raise NotImplementedError("Other adaptive projection methods are not implemented yet.")
        tgt_len, bsz, embed_dim = query.shape
        assert embed_dim == self.embed_dim, "query dim %d != %d" % (embed_dim, self.embed_dim)
        w, e, h, d = self.window_size, self.ext_size, self.num_heads, self.head_dim

        # Everything stays in fairseq's time-first layout: the kernels address q/k/v/out through
        # strides, so the batch-first views below cost no transposes of the activations.
        def padded(t):
            n_pad = int(math.ceil(tgt_len / w) * w) - tgt_len if w > 0 else 0
            return F.pad(t, (0, 0, 0, 0, 0, n_pad)) if n_pad else t

        x = padded(query)
        N, B, C = x.shape
        mask = None
        if key_padding_mask is not None or N != tgt_len:
            mask = torch.zeros(B, N, dtype=torch.bool, device=x.device)
            if key_padding_mask is not None:
                mask[:, :tgt_len] = key_padding_mask.to(torch.bool)
            mask[:, tgt_len:] = True
        y = self._forward_module(x, mask, N, B, C) if self.self_attention else None
        if y is not None:
            if N != tgt_len:
                y = y[:tgt_len]
            return y.contiguous(), None
        # (the full-sequence path has fp32 cores; decoding does not -- an explicit argument, not module state: forward stays
        #  re-entrant)
        if self.self_attention:
            qkv5 = self._project(x, None, None, keep_f32=True)
        else:
            qkv5 = self._project(x, padded(key), padded(value), keep_f32=True)
        qkv5 = qkv5.transpose(0, 1)                       # [B, N, 3, h, d] view of the time-first buffer

        r = self.chunk_size if self.chunk_size is not None else int(N // self.num_chunks)
        if r >= N:
            raise NotImplementedError("a single chunk spanning the sequence (the reference's own "
                                      "branch for it, causal_eva.py:680-683, does not run either)")
        assert N % r == 0, "sequence length %d is not a multiple of the chunk size %d" % (N, r)
        L = N // r
        bias = None
        # (the 16-bit kernel branch below takes the table itself: decided here so that the dense bias is not built twice)
        tb_ok = bool(self.use_t5_rpe and _ops.USE_TABLE_BIAS and x.is_cuda and qkv5.dtype != torch.float32
                     and not (L > 64 and _f32.ENABLED) and _ops._DIRECT and not torch.compiler.is_compiling()
                     and torch._C._len_torch_dispatch_stack() == 0)
        if self.use_t5_rpe and not tb_ok:
            bias = self.rel_pos_bias.dense(w, w + e, x.device).expand(h, w, w + e)
        noise = None
        if self.training:
            noise = torch.randn_like(torch.empty(B, h, L, d, device=x.device, dtype=torch.float32))
        def mu_fn(qm, km):
            rq, rk = self.adaptive_mu_q(qm), self.adaptive_mu_k(km)
            return rk, rq + rk
        if qkv5.dtype == torch.float32:
            # fp32 activations outside autocast (round 5): the core on the fp32-faithful kernels (causal_eva.py:666-783 in the
            # precision the reference computes it), the mu networks as the module's own layers
            out = _f32.causal_eva_core(qkv5, bias, noise, _ops._mask_u8(mask, B, N, x.device), w, e, r, bool(self.causal), mu_fn,
                                       *self._dropout_keep(B, h, N, w + e, L, x.device, raw=True))
        elif L > 64 and _f32.ENABLED:
            # more chunks than the 16-bit window kernels hold landmark rows (64; the recipe's 512-token samples have exactly
            # 64): the generic fp32 kernels on the 16-bit activations -- the reference takes any length (causal_eva.py:680-700)
            with torch.autocast(device_type="cuda", enabled=False):
                out = _f32.causal_eva_core(qkv5.float(), None if bias is None else bias.float(), noise,
                                           _ops._mask_u8(mask, B, N, x.device), w, e, r, bool(self.causal), mu_fn,
                                           *self._dropout_keep(B, h, N, w + e, L, x.device, raw=True)).to(qkv5.dtype)
        else:
            cfg = (False, (N,), w, e, r, L, "default" if self.adaptive_proj == "qk" else "no-ln",
                   2 if self.causal else 1, 1.0) + (self._dropout_keep(B, h, N, w + e, L, x.device) or (None, 1.0))
            if tb_ok:
                # round 6: the single-head T5 table handed over as it is -- the [h, w, w + e] bias built from it, and its
                # gradient taken back to it, by one launch each way inside the core's node (_ops.TableBias)
                bias = self.rel_pos_bias.relative_attention_bias.weight
                cfg = cfg + (self.rel_pos_bias.table_spec(w, w + e),)
            out = _ops.EvaAttnFn.apply(qkv5, bias, noise, _ops._mask_u8(mask, B, N, x.device), cfg,
                                       *self._mu_params())
        # out [B, N, h, d] comes back as a view of a time-first buffer (it follows qkv's layout)
        self._quant_noise_(self.out_proj)
        y = _ops.linear(out.transpose(0, 1).reshape(N, B, C), self.out_proj)
        if not torch.is_autocast_enabled() and y.dtype != query.dtype:
            y = y.to(query.dtype)
        if N != tgt_len:
            y = y[:tgt_len]
        return y.contiguous(), None

    # ---- the training step as ONE autograd node ---------------------------------------------
    def _forward_module(self, x, mask, N, B, C):
        """Self-attention under 16-bit autocast with autograd on (a training / scoring-with-gradients step): q / k / v projection ->
        causal EVA core -> output projection as one autograd node (_ops.CoreModuleFn around an _ops.GraphCore that records the
        core's own Function), so that both projections' weight gradients leave in one launch and their partial sums in one
        more -- the three-node path below pays two of each.  Returns None when it does not apply (fp32 activations, more than
        64 chunks, quantization noise in training -- its draws keep the reference's order on the three-node path --, tracing)."""
        w, e, h, d = self.window_size, self.ext_size, self.num_heads, self.head_dim
        if not (_ops.USE_CAUSAL_MODULE_FN and torch.is_autocast_enabled() and torch.is_grad_enabled() and x.is_cuda
                and not (self.training and self.q_noise > 0)):
            return None
        cdtype = torch.get_autocast_dtype("cuda")
        r = self.chunk_size if self.chunk_size is not None else int(N // max(int(self.num_chunks or 1), 1))
        if cdtype not in (torch.bfloat16, torch.float16) or r <= 0 or r >= N or N % r != 0 or N // r > 64:
            return None
        L = N // r
        ws = [self.q_proj.weight, self.k_proj.weight, self.v_proj.weight]
        bs = [self.q_proj.bias, self.k_proj.bias, self.v_proj.bias]
        if any(wt.dtype != torch.float32 or not wt.is_contiguous() for wt in ws) or any((b is None) != (bs[0] is None) for b in bs):
            return None
        wq = torch.cat(ws, 0)
        bq = None if bs[0] is None else torch.cat(bs, 0)

        class _W:                                        # (what core_module_fn_supported reads of an nn.Linear)
            def __init__(self, weight):
                self.weight = weight
        if not _ops.core_module_fn_supported(x, _W(wq), self.out_proj, cdtype):
            return None
        tb_ok = bool(self.use_t5_rpe and _ops.USE_TABLE_BIAS)
        noise = None
        if self.training:
            noise = torch.randn_like(torch.empty(B, h, L, d, device=x.device, dtype=torch.float32))
        mask_u8 = _ops._mask_u8(mask, B, N, x.device)
        mu = self._mu_params()
        table = self.rel_pos_bias.relative_attention_bias.weight if self.use_t5_rpe else None
        adaptive = "default" if self.adaptive_proj == "qk" else "no-ln"

        def core(qkv_tf, *params):                       # [N, B, 3, h, d] time-first -> out [N, B, h, d]
            qkv5 = qkv_tf.transpose(0, 1)
            cfg = (False, (N,), w, e, r, L, adaptive, 2 if self.causal else 1, 1.0) \
                + (self._dropout_keep(B, h, N, w + e, L, x.device) or (None, 1.0))
            bias = None
            if tb_ok:
                bias = table
                cfg = cfg + (self.rel_pos_bias.table_spec(w, w + e),)
            elif self.use_t5_rpe:
                bias = self.rel_pos_bias.dense(w, w + e, x.device).expand(h, w, w + e)
            out = _ops.EvaAttnFn.apply(qkv5, bias, noise, mask_u8, cfg, *mu)
            return out.transpose(0, 1)

        inputs = ([table] if table is not None else []) + list(mu)
        y = _ops.CoreModuleFn.apply(x, wq, bq, self.out_proj.weight, self.out_proj.bias,
                                    _ops.GraphCore(core, len(inputs), True), cdtype, h, *inputs)
        return y

    # ---- incremental decoding (reference :537-665) ------------------------------------------
    def _decode(self, query, key_padding_mask, incremental_state):
        """Token-by-token decoding with fairseq's incremental state.

        The reference's branch for this (causal_eva.py:537-665) cannot run as shipped -- `N` and `B` are bound only when
        `incremental_state is None` (:503-509), so any call with a state raises -- and what it sketches (a sliding window of
        the last `window_size` keys) would not agree with the module's own training path (block windows with a left
        extension).  This build therefore defines decoding by PREFIX CONSISTENCY with the pinned full-sequence path: the
        output for token t equals row t of `forward()` on the tokens 0..t (tests/test_gpu_causal_eva.py).  State, all
        batch-first so that `reorder_incremental_state` can index it:
            qkv       [B, cap, 3, h, d]   projected rows of every token so far (16-bit; the window needs the last w + e,
                                          a chunk its own rows)
            rf_k_bar  [B, h, Lcap, d]     fp32, the landmark keys of the COMPLETED chunks (:588-634)
            beta      [B, h, Lcap, d]     fp32, their control variates
            pos       [B]                 tokens decoded so far
            pad       [B, cap]            uint8, 1 = padded position (`key_padding_mask`, e.g. left-padded prompts of a batch):
                                          handed to the kernels exactly as the full path hands them its mask -- a padded key
                                          is invisible, a padded query sees no local key, a chunk's means skip its padded rows
        A step projects the new token, closes a chunk when one completes (chunk means -> mu networks -> beta, the same HIP
        entry points as the full path on the chunk's rows) and runs the window kernel of the training path on the suffix
        [previous window, current window] against the landmarks of all completed chunks (`ea_geom.lm_base` re-bases the
        chunk visibility rule of :716-738); rows after the current token are masked.  Past 64 completed chunks (the landmark rows
        the 16-bit window kernel holds) the same suffix runs on the generic fp32 kernel: no length limit."""
        if not self.self_attention:
            raise NotImplementedError("incremental decoding of encoder-decoder attention")
        if not self.causal:
            raise NotImplementedError("incremental decoding needs --causal: without the causal masks every query of the "
                                      "training path sees the landmarks of future chunks (causal_eva.py:716-738)")
        if self.training:
            raise NotImplementedError("incremental decoding in training mode")
        if self.adaptive_proj not in ("qk", "no-ln"):
            raise NotImplementedError("Other adaptive projection methods are not implemented yet.")
        _ops.nv.require_cuda(query, "query")                       # (before any state is built: no CPU fallback)
        T_new, B, C = query.shape
        if key_padding_mask is not None:
            # fairseq hands the decoder either the flags of the new positions [B, T_new] or of every position so far
            # [B, t0 + T_new] (`self_attn_padding_mask`): the last T_new columns are this step's in both cases
            if key_padding_mask.dim() != 2 or key_padding_mask.shape[0] != B or key_padding_mask.shape[1] < T_new:
                raise ValueError("key_padding_mask %s does not cover the %d new positions of a batch of %d"
                                 % (tuple(key_padding_mask.shape), T_new, B))
        w, e, h, d = self.window_size, self.ext_size, self.num_heads, self.head_dim
        r = self.chunk_size
        if r is None:
            raise NotImplementedError("incremental decoding needs --chunk-size (with --num-chunks the chunk length "
                                      "depends on the final sequence length)")
        dev = query.device
        state = self._get_input_buffer(incremental_state)
        qkv_new = self._project(query, None, None)                 # [T_new, B, 3, h, d]
        if "qkv" not in state:
            cap = max(2 * w, 64)
            state["qkv"] = torch.zeros((B, cap, 3, h, d), dtype=qkv_new.dtype, device=dev)
            lcap = max(cap // r, 1)
            state["rf_k_bar"] = torch.zeros((B, h, lcap, d), dtype=torch.float32, device=dev)
            state["beta"] = torch.zeros((B, h, lcap, d), dtype=torch.float32, device=dev)
            state["pos"] = torch.zeros((B,), dtype=torch.long, device=dev)
            state["pad"] = torch.zeros((B, cap), dtype=torch.uint8, device=dev)
            self.set_incremental_state(incremental_state, "attn_pos", 0)
            self.set_incremental_state(incremental_state, "attn_has_pad", False)
        # (the token count also lives on the host, under its own key of the incremental state -- reading `pos` back would
        #  synchronise every step, and reorder_incremental_state only touches the tensors of the buffer)
        t0 = int(self.get_incremental_state(incremental_state, "attn_pos") or 0)
        if state["qkv"].shape[0] != B:
            raise RuntimeError("incremental state holds batch %d, the step has %d" % (state["qkv"].shape[0], B))
        need = ((t0 + T_new + w - 1) // w) * w
        if need > state["qkv"].shape[1]:
            cap = max(need, 2 * state["qkv"].shape[1])
            grown = torch.zeros((B, cap, 3, h, d), dtype=state["qkv"].dtype, device=dev)
            grown[:, :state["qkv"].shape[1]] = state["qkv"]
            state["qkv"] = grown
            gpad = torch.zeros((B, cap), dtype=torch.uint8, device=dev)
            gpad[:, :state["pad"].shape[1]] = state["pad"]
            state["pad"] = gpad
            lcap = cap // r
            for name in ("rf_k_bar", "beta"):
                g2 = torch.zeros((B, h, lcap, d), dtype=torch.float32, device=dev)
                g2[:, :, :state[name].shape[2]] = state[name]
                state[name] = g2
        cache = state["qkv"]
        cache[:, t0:t0 + T_new] = qkv_new.transpose(0, 1)
        # (whether a mask was ever given lives on the host: the unpadded case keeps its mask-free chunk kernels without
        #  reading a flag back from the device)
        has_pad = bool(self.get_incremental_state(incremental_state, "attn_has_pad"))
        if key_padding_mask is not None:
            state["pad"][:, t0:t0 + T_new] = key_padding_mask[:, -T_new:].to(device=dev, dtype=torch.uint8)
            if not has_pad:
                has_pad = True
                self.set_incremental_state(incremental_state, "attn_has_pad", True)
        pad = state["pad"]
        bias = None
        if self.use_t5_rpe:
            bias = self.rel_pos_bias.dense(w, w + e, dev).expand(h, w, w + e)
        mlp = self._mu_params()
        adaptive = "default" if self.adaptive_proj == "qk" else "no-ln"
        io = _ops.nv.io_dtype(cache)
        outs = []
        for i in range(T_new):
            t = t0 + i
            # ---- landmarks of the chunks before this token's own (completed at the previous steps) ----
            nvis = t // r
            b = t // w
            b0 = max(b - 1, 0) if e > 0 else b
            Nc = (b - b0 + 1) * w
            ctx = cache[:, b0 * w:b0 * w + Nc]                      # [B, Nc, 3, h, d] view
            mask = pad[:, b0 * w:b0 * w + Nc].clone() if has_pad else torch.zeros((B, Nc), dtype=torch.uint8, device=dev)
            mask[:, t - b0 * w + 1:] = 1                            # rows after the current token: not decoded yet
            lk = state["rf_k_bar"][:, :, :nvis].contiguous() if nvis else None
            lv = state["beta"][:, :, :nvis].contiguous() if nvis else None
            if nvis <= 64:
                geom = _ops.nv.make_geom(B, h, Nc, d, io, False, (Nc,), w, e, r, nvis, 2, (b0 * w) // r)
                bias_p = _ops._bias_padded(bias, geom)
                out, _ = _ops._window_fwd(geom, ctx, lk, lv, bias_p, mask)
                outs.append(out[:, t - b0 * w])                     # [B, h, d]
            else:
                # more completed chunks than the window kernel holds landmark rows: the same suffix on the generic fp32
                # kernel (exact on the 16-bit rows; `lm_base` re-bases the visibility rule exactly as ea_geom's does)
                q32, k32, v32 = [ctx[:, :, i].float().permute(0, 2, 1, 3) for i in range(3)]
                spec = dict(idx_q=_f32.window_table_1d(Nc, w, 0, dev), idx_k=_f32.window_table_1d(Nc, w, e, dev, left_only=True),
                            kmask=mask, qmask=mask, scale=d ** -0.5, causal_e=e, chunk=r, lm_base=(b0 * w) // r)
                with torch.autocast(device_type="cuda", enabled=False):
                    out, _ = _f32.GatherAttnFn.apply(q32, k32, v32, lk, lv, None if bias is None else bias.float(), spec)
                outs.append(out[:, :, t - b0 * w].to(cache.dtype))  # [B, h, d]
            # ---- this token closes chunk c: its landmark becomes visible from the next chunk on ----
            if (t + 1) % r == 0:
                c = t // r
                sub = cache[:, c * r:(c + 1) * r]
                icfg = [0, r, 0, r, 0, r, 1, 1, 0]
                cmask = pad[:, c * r:(c + 1) * r].contiguous() if has_pad else None
                res = torch.ops.ea.eva_fwd(sub, None, None, cmask, None, icfg, [1.0, 1.0], adaptive, list(mlp))
                state["beta"][:, :, c] = res[6][:, :, 0]
                state["rf_k_bar"][:, :, c] = res[7][:, :, 0]
        self.set_incremental_state(incremental_state, "attn_pos", t0 + T_new)
        state["pos"] = state["pos"] + T_new
        self._set_input_buffer(incremental_state, state)
        o = torch.stack(outs, 0).reshape(T_new, B, C)               # [T_new, B, h*d]
        y = _ops.linear(o, self.out_proj)
        if not torch.is_autocast_enabled() and y.dtype != query.dtype:
            y = y.to(query.dtype)
        return y.contiguous(), None

    def _dropout_keep(self, B, h, N, Wk, L, device, raw=False):
        """Attention dropout (reference :778, `attn = dropout(attn)` on the [.., Wk + L] softmax rows):
        the Bernoulli keep decisions are drawn here -- one per (query, column), the reference's
        layout -- and handed to the kernels as a uint8 mask; -> (keep, 1/(1-p)) or ()."""
        p = self.dropout_module.p
        if not (self.training and p > 0):
            return ()
        if p >= 1:
            raise NotImplementedError("attention dropout with p = 1")
        if self._keep_mask_fn is not None:                         # tests: the fixture's decisions
            keep = self._keep_mask_fn((B, h, N, Wk + L)).to(device=device, dtype=torch.uint8)
        else:
            keep = torch.empty((B, h, N, Wk + L), device=device, dtype=torch.uint8).bernoulli_(1 - p)
        if raw:                                                    # the reference's own layout (the fp32 cores take it as is)
            return (keep.contiguous(), 1.0 / (1.0 - p))
        # kernel layout: local columns padded to whole 16-key tiles, then the landmarks
        ld_local, ld_lm = -(-Wk // 16) * 16, -(-L // 16) * 16
        if ld_local != Wk or ld_lm != L:
            padded = torch.zeros((B, h, N, ld_local + ld_lm), device=device, dtype=torch.uint8)
            padded[..., :Wk] = keep[..., :Wk]
            padded[..., ld_local:ld_local + L] = keep[..., Wk:]
            keep = padded
        return (keep.contiguous(), 1.0 / (1.0 - p))

    # ---- fairseq incremental-state protocol (reference :262-297, 836-871) -------------------
    def init_incremental_state(self):
        self._incremental_state_id = str(uuid.uuid4())

    def _get_full_incremental_state_key(self, key):
        return "%s.%s" % (self._incremental_state_id, key)

    def get_incremental_state(self, incremental_state, key):
        full = self._get_full_incremental_state_key(key)
        if incremental_state is None or full not in incremental_state:
            return None
        return incremental_state[full]

    def set_incremental_state(self, incremental_state, key, value):
        if incremental_state is not None:
            incremental_state[self._get_full_incremental_state_key(key)] = value
        return incremental_state

    def _get_input_buffer(self, incremental_state):
        found = self.get_incremental_state(incremental_state, "attn_state")
        return {} if found is None else found

    def _set_input_buffer(self, incremental_state, buffer):
        return self.set_incremental_state(incremental_state, "attn_state", buffer)

    def reorder_incremental_state(self, incremental_state, new_order):
        buf = self._get_input_buffer(incremental_state)
        if buf:
            for k, t in buf.items():
                if t is not None:
                    buf[k] = t.index_select(0, new_order)
            incremental_state = self._set_input_buffer(incremental_state, buf)
        return incremental_state

    def apply_sparse_mask(self, attn_weights, tgt_len, src_len, bsz):
        return attn_weights

    def upgrade_state_dict_named(self, state_dict, name):
        """Split a legacy fused `in_proj_weight/bias` into q/k/v projections (reference :876-903)."""
        prefix = name + "." if name != "" else ""
        for key in [k for k in state_dict if k.endswith(prefix + "in_proj_weight")]:
            fused = state_dict.pop(key)
            dim = fused.shape[0] // 3
            for i, p in enumerate(("q_proj", "k_proj", "v_proj")):
                state_dict[prefix + p + ".weight"] = fused[i * dim:(i + 1) * dim]
            bkey = prefix + "in_proj_bias"
            if bkey in state_dict:
                fb = state_dict.pop(bkey)
                for i, p in enumerate(("q_proj", "k_proj", "v_proj")):
                    state_dict[prefix + p + ".bias"] = fb[i * dim:(i + 1) * dim]

    @staticmethod
    def add_attn_specific_args(parent_parser, struct_name="attn_args", prefix=""):
        group = parent_parser.add_argument_group("attention")
        fp = prefix + "-" if len(prefix) > 1 else ""
        kw = dict(struct_name=struct_name, prefix=prefix)
        add_nested_argument(group, "--%sadaptive-proj" % fp, default="default", type=str, **kw)
        add_nested_argument(group, "--%snum-chunks" % fp, default=None, type=int, **kw)
        add_nested_argument(group, "--%schunk-size" % fp, default=None, type=int, **kw)
        add_nested_argument(group, "--%scausal" % fp, action="store_true", default=False, **kw)
        add_nested_argument(group, "--%suse-t5-rpe" % fp, action="store_true", default=False, **kw)
        add_nested_argument(group, "--%swindow-size" % fp, default=4, type=int, **kw)
        add_nested_argument(group, "--%soverlap-window" % fp, action="store_true", default=False, **kw)
        return parent_parser
