"""torch.ops.ea.* -- the attention cores as dispatcher-registered custom ops (last bullet).

One forward and one backward op per variant, with explicit schemas (tensors, Tensor? for absent
operands, int[] / float[] geometry) and fake-tensor implementations, so the cores are visible to the
dispatcher (torch.ops.ea.<name>), to torch.compile as opaque nodes and to anything that walks a graph.
Each op's CUDA implementation is the functional form in _ops.py, i.e. a short sequence of C-ABI launches
on the current stream (include/ea_hip.h); there is no CPU kernel -- a CPU tensor raises in the
implementation (no fallback).  The autograd Functions of _ops.py are thin shells over these ops.

    ea::softmax_fwd / softmax_bwd        abstract_attention.py:120-133
    ea::local_fwd / local_bwd            local_attention.py:134-182
    ea::eva_fwd / eva_bwd                eva.py:145-227 (and causal_eva.py:666-788)
    ea::lara_fwd / lara_bwd              lara.py:129-175,187-246 (2-D pooled proposals)
    ea::performer_fwd / performer_bwd    kernelized_attention.py:20-56,116-121
    ea::linear / linear_w32              abstract_attention.py:72-78,86-87 (qkv / output projection, streaming kernel;
                                         _w32: straight from the fp32 master weight)
"""
import torch

from . import _ops

_LIB = torch.library.Library("ea", "DEF")

_SCHEMAS = {
    "softmax_fwd": "(Tensor qkv, Tensor? mask, Tensor? keep, float keep_scale) -> Tensor[]",
    "softmax_bwd": "(Tensor dout, Tensor qkv, Tensor? mask, Tensor out, Tensor lse, Tensor? keep, float keep_scale) -> Tensor",
    "local_fwd": "(Tensor qkv, Tensor? bias, Tensor? mask, int[] geo) -> Tensor[]",
    "local_bwd": "(Tensor dout, Tensor? dlse, Tensor qkv, Tensor? bias_p, Tensor? mask, Tensor out, Tensor lse, "
                 "int[] geo, int bias_cols) -> Tensor[]",
    "performer_fwd": "(Tensor qkv, Tensor? mask, Tensor W) -> Tensor[]",
    "performer_bwd": "(Tensor dout, Tensor qkv, Tensor? mask, Tensor W, Tensor stab, Tensor kv, Tensor ksum, "
                     "Tensor out) -> Tensor",
    "lara_fwd": "(Tensor qkv, Tensor? mask, Tensor? noise, int[] icfg, float[] fcfg, Tensor[] params) -> Tensor[]",
    "lara_bwd": "(Tensor dout, Tensor qkv, Tensor? mask, Tensor? noise, Tensor[] saved, int[] icfg, float[] fcfg, Tensor[] params) "
                "-> Tensor[]",
    "eva_fwd": "(Tensor qkv, Tensor? bias, Tensor? noise, Tensor? mask, Tensor? keep, int[] icfg, float[] fcfg, "
               "str adaptive_proj, Tensor[] params) -> Tensor[]",
    "eva_bwd": "(Tensor dout, Tensor qkv, Tensor? mask, Tensor? keep, Tensor? noise, Tensor out, Tensor[] saved, int[] icfg, float[] fcfg, "
               "str adaptive_proj, int bias_cols, Tensor[] params) -> Tensor[]",
    "linear": "(Tensor a, Tensor w, Tensor? bias, bool y_f32, bool want_cast) -> Tensor[]",
    "linear_w32": "(Tensor a, Tensor w32, Tensor? bias, int elem, bool transposed, bool y_f32, bool want_cast) -> Tensor[]",
}
_IMPLS = {
    "softmax_fwd": _ops.softmax_fwd_impl, "softmax_bwd": _ops.softmax_bwd_impl,
    "local_fwd": _ops.local_fwd_impl, "local_bwd": _ops.local_bwd_impl,
    "performer_fwd": _ops.performer_fwd_impl, "performer_bwd": _ops.performer_bwd_impl,
    "lara_fwd": _ops.lara_fwd_impl, "lara_bwd": _ops.lara_bwd_impl,
    "eva_fwd": _ops.eva_fwd_impl, "eva_bwd": _ops.eva_bwd_impl,
    "linear": _ops.linear_impl, "linear_w32": _ops.linear_w32_impl,
}
def _no_cpu(*args, **kwargs):
    _ops.nv.require_cuda(None, "every tensor of torch.ops.ea.*")       # raises: the cores have no CPU fallback


for _name, _schema in _SCHEMAS.items():
    _LIB.define(_name + _schema)
    _LIB.impl(_name, _IMPLS[_name], "CUDA")
    _LIB.impl(_name, _no_cpu, "CPU")


def _f32(like, *shape):
    return like.new_empty(shape, dtype=torch.float32)


def _none(like):
    return like.new_empty(0, dtype=torch.float32)


@torch.library.register_fake("ea::linear")
def _(a, w, bias, y_f32, want_cast):
    y = a.new_empty((a.shape[0], w.shape[0]), dtype=torch.float32 if y_f32 else w.dtype)
    ac = a.new_empty(a.shape if (want_cast and a.dtype == torch.float32) else (0,), dtype=w.dtype)
    return [y, ac]


@torch.library.register_fake("ea::linear_w32")
def _(a, w32, bias, elem, transposed, y_f32, want_cast):
    dt = _ops._ELEM_DT[int(elem)]
    y = a.new_empty((a.shape[0], w32.shape[1] if transposed else w32.shape[0]), dtype=torch.float32 if y_f32 else dt)
    ac = a.new_empty(a.shape if (want_cast and a.dtype == torch.float32) else (0,), dtype=dt)
    return [y, ac]


@torch.library.register_fake("ea::softmax_fwd")
def _(qkv, mask, keep, keep_scale):
    B, N, _, h, d = qkv.shape
    return [qkv.new_empty((B, N, h, d)), _f32(qkv, B * h, N)]


@torch.library.register_fake("ea::softmax_bwd")
def _(dout, qkv, mask, out, lse, keep, keep_scale):
    return torch.empty_like(qkv)


@torch.library.register_fake("ea::local_fwd")
def _(qkv, bias, mask, geo):
    B, N, _, h, d = qkv.shape
    bias_p = _none(qkv) if bias is None else _f32(qkv, bias.shape[0], bias.shape[1], -(-bias.shape[2] // 16) * 16)
    return [qkv.new_empty((B, N, h, d)), _f32(qkv, B, h, N), bias_p]


@torch.library.register_fake("ea::local_bwd")
def _(dout, dlse, qkv, bias_p, mask, out, lse, geo, bias_cols):
    dbias = _none(qkv) if bias_p is None else _f32(qkv, bias_p.shape[0], bias_p.shape[1], bias_cols)
    return [torch.empty_like(qkv), dbias]


@torch.library.register_fake("ea::performer_fwd")
def _(qkv, mask, W):
    B, N, _, h, d = qkv.shape
    m = W.shape[1]
    return [qkv.new_empty((B, N, h, d)), _f32(qkv, B * h), _f32(qkv, B * h, m, d), _f32(qkv, B * h, m)]


@torch.library.register_fake("ea::performer_bwd")
def _(dout, qkv, mask, W, stab, kv, ksum, out):
    return torch.empty_like(qkv)


@torch.library.register_fake("ea::lara_fwd")
def _(qkv, mask, noise, icfg, fcfg, params):
    B, N, _, h, d = qkv.shape
    H, W, r, has_mlp, mixed, mis, dup = [int(v) for v in icfg[:7]]
    L = (H // r) * (W // r)
    C = L * (2 if dup else 1)
    BH = B * h
    # (host-side size query: works on fake tensors; the composite decision is taken like the real forward takes it -- the
    # traced output count must be what the implementation will return when the graph runs)
    lcfg, sizes = _ops._lara_layer_cfg(qkv, icfg, fcfg) if _ops._lara_use_composite() else (None, None)
    if lcfg is not None:
        return [qkv.new_empty((B, N, h, d)), _f32(qkv, sizes[0])]
    return [qkv.new_empty((B, N, h, d)), _f32(qkv, BH, C, d), _f32(qkv, BH, C, d) if mis != 2 else _none(qkv),
            _f32(qkv, BH, C) if mis == 0 else _none(qkv), _f32(qkv, BH, C), _f32(qkv, BH, C, d), _f32(qkv, BH, C),
            _f32(qkv, BH, C) if mis == 0 else _none(qkv), _f32(qkv, BH, L, d), _f32(qkv, BH, L, d),
            _f32(qkv, BH * (3 * L * d + L * 64 + 128)) if (len(icfg) < 8 or icfg[7]) else _none(qkv),
            _f32(qkv, 2, BH, N) if ((len(icfg) < 8 or icfg[7]) and C <= 64) else _none(qkv)]


@torch.library.register_fake("ea::lara_bwd")
def _(dout, qkv, mask, noise, saved, icfg, fcfg, params):
    return [torch.empty_like(qkv)] + [torch.empty_like(p, dtype=torch.float32) for p in params]


@torch.library.register_fake("ea::eva_fwd")
def _(qkv, bias, noise, mask, keep, icfg, fcfg, adaptive_proj, params):
    B, N, _, h, d = qkv.shape
    L = int(icfg[6])
    lm = _f32(qkv, B, h, L, d)
    bias_p = _none(qkv) if bias is None else _f32(qkv, bias.shape[0], bias.shape[1], -(-bias.shape[2] // 16) * 16)
    fused = adaptive_proj == "default" and L <= 64 and d in (32, 64) and float(fcfg[0]) == 0.5
    need = len(icfg) < 9 or bool(icfg[8])
    saved = _f32(qkv, B * h * (3 * L * d + L * 64 + 128)) if (fused and need) else _none(qkv)
    sides = 1 if adaptive_proj == "none" else 2
    ln = (not fused) and need and adaptive_proj != "no-ln"
    zhat = _f32(qkv, sides, B * h * L, d) if ln else _none(qkv)
    rstd = _f32(qkv, sides, B * h * L) if ln else _none(qkv)
    return [qkv.new_empty((B, N, h, d)), bias_p, _f32(qkv, B, h, N), lm, torch.empty_like(lm), torch.empty_like(lm),
            torch.empty_like(lm), torch.empty_like(lm), saved, zhat, rstd]


@torch.library.register_fake("ea::eva_bwd")
def _(dout, qkv, mask, keep, noise, out, saved, icfg, fcfg, adaptive_proj, bias_cols, params):
    bias_p = saved[0]
    dbias = _none(qkv) if bias_p.numel() == 0 else _f32(qkv, bias_p.shape[0], bias_p.shape[1], bias_cols)
    return [torch.empty_like(qkv), dbias] + [torch.empty_like(p, dtype=torch.float32) for p in params]
