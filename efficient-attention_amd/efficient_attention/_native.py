"""ctypes binding of libea_hip.so (include/ea_hip.h) -- the only way the attention cores run.

There is NO CPU or PyTorch fallback in this package: if the library cannot be loaded, or a
tensor that is not a contiguous-enough CUDA/HIP tensor reaches a core, a RuntimeError is raised.
torch is used here for device memory and streams only (`data_ptr()`, current stream handle).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get(
    "EA_HIP_LIB", os.path.join(os.path.dirname(_HERE), "lib", "libea_hip.so"))

EA_BF16, EA_F16, EA_F32 = 0, 1, 2
_DTYPES = {torch.bfloat16: EA_BF16, torch.float16: EA_F16}


class ea_t4(ctypes.Structure):
    _fields_ = [("ptr", ctypes.c_void_p), ("sb", ctypes.c_int64), ("sh", ctypes.c_int64),
                ("sn", ctypes.c_int64)]


ABI_VERSION = 15         # ea_abi_version() of include/ea_hip.h this file mirrors


class ea_geom(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("H", ctypes.c_int32), ("N", ctypes.c_int32),
                ("D", ctypes.c_int32), ("dtype", ctypes.c_int32), ("attn_2d", ctypes.c_int32),
                ("gh", ctypes.c_int32), ("gw", ctypes.c_int32), ("window", ctypes.c_int32),
                ("ext", ctypes.c_int32), ("chunk", ctypes.c_int32), ("L", ctypes.c_int32),
                ("scale", ctypes.c_float), ("causal", ctypes.c_int32), ("lm_base", ctypes.c_int32)]


class ea_perf_geom(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("H", ctypes.c_int32), ("N", ctypes.c_int32),
                ("D", ctypes.c_int32), ("dtype", ctypes.c_int32), ("M", ctypes.c_int32)]


class ea_sb_geom(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("H", ctypes.c_int32), ("N", ctypes.c_int32), ("D", ctypes.c_int32),
                ("dtype", ctypes.c_int32), ("M", ctypes.c_int32), ("attn_2d", ctypes.c_int32),
                ("gh", ctypes.c_int32), ("gw", ctypes.c_int32), ("window", ctypes.c_int32)]


class ea_lmk_geom(ctypes.Structure):
    _fields_ = [("BH", ctypes.c_int32), ("L", ctypes.c_int32), ("C", ctypes.c_int32), ("D", ctypes.c_int32),
                ("has_mlp", ctypes.c_int32), ("mixed", ctypes.c_int32), ("mis", ctypes.c_int32),
                ("dup", ctypes.c_int32), ("scale", ctypes.c_float), ("eva", ctypes.c_int32)]


class ea_lara_layer(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("H", ctypes.c_int32), ("D", ctypes.c_int32), ("dtype", ctypes.c_int32),
                ("gh", ctypes.c_int32), ("gw", ctypes.c_int32), ("pool_r", ctypes.c_int32),
                ("has_mlp", ctypes.c_int32), ("mixed", ctypes.c_int32), ("mis", ctypes.c_int32), ("dup", ctypes.c_int32),
                ("kappa", ctypes.c_float), ("scale", ctypes.c_float)]


class ea_f32_attn(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("H", ctypes.c_int32), ("Nq", ctypes.c_int32), ("Nk", ctypes.c_int32), ("D", ctypes.c_int32),
                ("G", ctypes.c_int32), ("Wq", ctypes.c_int32), ("Wk", ctypes.c_int32), ("L", ctypes.c_int32),
                ("knorm", ctypes.c_int32), ("neg_inf", ctypes.c_int32), ("causal_e", ctypes.c_int32), ("chunk", ctypes.c_int32),
                ("lm_base", ctypes.c_int32), ("bias_ld", ctypes.c_int32), ("bias_hs", ctypes.c_int64), ("bias_bs", ctypes.c_int64),
                ("keep_ld", ctypes.c_int64),
                ("keep_scale", ctypes.c_float), ("scale", ctypes.c_float)]


class ea_eva_layer(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("H", ctypes.c_int32), ("D", ctypes.c_int32), ("dtype", ctypes.c_int32),
                ("gh", ctypes.c_int32), ("gw", ctypes.c_int32), ("window", ctypes.c_int32), ("chunk", ctypes.c_int32),
                ("has_bias", ctypes.c_int32), ("scale", ctypes.c_float)]


class ea_lara_geom(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("H", ctypes.c_int32), ("N", ctypes.c_int32),
                ("D", ctypes.c_int32), ("dtype", ctypes.c_int32), ("C", ctypes.c_int32),
                ("mis", ctypes.c_int32), ("kappa", ctypes.c_float), ("scale", ctypes.c_float)]


_P = ctypes.c_void_p
_I = ctypes.c_int32
_L = ctypes.c_int64
_F = ctypes.c_float
_G = ctypes.POINTER(ea_geom)
_LG = ctypes.POINTER(ea_lara_geom)
_PG = ctypes.POINTER(ea_perf_geom)
_MG = ctypes.POINTER(ea_lmk_geom)
_T = ctypes.POINTER(ea_t4)
_SG = ctypes.POINTER(ea_sb_geom)
_LL = ctypes.POINTER(ea_lara_layer)
_EL = ctypes.POINTER(ea_eva_layer)
_FA = ctypes.POINTER(ea_f32_attn)

# name -> argtypes; every symbol include/ea_hip.h declares (tests check the list is complete)
SIGNATURES = {
    "ea_eva_chunk_mean_fwd": [_G, _T, _T, _P, _P, _P, _P],
    "ea_eva_chunk_mean_bwd": [_G, _P, _P, _P, _T, _T, _P],
    "ea_eva_beta_fwd": [_G, _T, _T, _P, _P, _P, _P],
    "ea_eva_beta_bwd": [_G, _T, _T, _P, _P, _P, _P, _T, _T, _P, _P],
    "ea_window_attn_fwd": [_G, _T, _T, _T, _P, _P, _P, _P, _T, _P, _P, _F, _P],
    "ea_window_attn_bwd": [_G, _T, _T, _T, _P, _P, _P, _P, _T, _T, _P, _T, _T, _T, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P],
    "ea_window_keep_ld": [_G],
    "ea_rows_mlp_parts": [_I, _I],
    "ea_rows_mlp_fwd": [_I] * 4 + [_P] * 15,
    "ea_rows_mlp_bwd": [_I] * 4 + [_P] * 15,
    "ea_lara_landmarks_fwd": [_MG] + [_P] * 17,
    "ea_lara_landmarks_bwd": [_MG] + [_P] * 21,
    "ea_lara_landmarks_bwd_parts": [_MG] + [_P] * 12 + [_I, _P, _F] + [_P] * 9,
    "ea_lara_out_fwd_merge": [_LG, _T, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _T, _P, _P, _P],
    "ea_lara_bwd_k_fused_merge": [_LG, _T, _T, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _T, _T, _P, _P, _P, _P, _P, _P, _P],
    "ea_lara_landmarks_fwd_cb": [_MG] + [_P] * 18,
    "ea_lara_landmarks_bwd_cb": [_MG] + [_P] * 23,
    "ea_lara_landmarks_saved_floats": [_MG],
    "ea_lara_merge_fwd": [_I] * 5 + [_P] * 8,
    "ea_lara_merge_bwd": [_I] * 5 + [_F] + [_P] * 16,
    "ea_bias_grad_parts": [_I, _I],
    "ea_bias_grad": [_I, _I, _I, _P, _P, _P, _P],
    "ea_colsum_f32": [_I, _I, _P, _P, _P],
    "ea_colsum2_f32": [_I, _I, _P, _P, _I, _P, _P, _P],
    "ea_slice_sum": [_I, _I, _I, _F, _P, _P, _P, _P],
    "ea_stream_copy": [_P, _P, _L, _P],
    "ea_lara_segment_fwd": [_G, _T, _T] + [_P] * 12,
    "ea_lara_segment_bwd": [_G, _T, _T] + [_P] * 11 + [_T, _T, _P, _P],
    "ea_lara_parts": [_LG],
    "ea_lara_stats_fwd": [_LG, _T, _T, _T, _P, _P, _P, _P, _P, _P],
    "ea_lara_out_fwd": [_LG, _T, _P, _P, _P, _P, _P, _P, _T, _P, _P, _P],
    "ea_lara_bwd_q": [_LG, _T, _T, _P, _P, _P, _P, _P, _P, _T, _P, _P, _P, _P, _P],
    "ea_lara_bwd_qstats": [_LG, _T, _T] + [_P] * 16,
    "ea_lara_bwd_k": [_LG, _T, _T, _P, _P, _P, _P, _P, _P, _T, _T, _P],
    "ea_lara_bwd_kstats": [_LG, _T, _T] + [_P] * 8,
    "ea_lara_bwd_qcorr": [_LG, _T, _P, _P, _P, _T, _P],
    "ea_lara_fused_parts": [_LG],
    "ea_lara_bwd_q_fused": [_LG, _T, _T, _P, _P, _P, _P, _P, _P, _P, _P, _T, _P, _P, _P, _P, _P, _P],
    "ea_lara_bwd_k_fused": [_LG, _T, _T, _P, _P, _P, _P, _P, _P, _T, _T, _P, _P],
    "ea_lara_bwd_finish": [_LG, _T, _P, _P, _P, _P, _P, _I, _I, _I, _T, _T, _P],
    "ea_lara_sample_fwd": [_I, _I, _I, _I, _I, _I, _F] + [_P] * 8,
    "ea_lara_sample_bwd": [_I, _I, _I, _I, _I, _I, _F] + [_P] * 10,
    "ea_adaptive_pool2d_fwd": [_I, _I, _I, _I, _I, _I, _I, _T, _P, _P],
    "ea_adaptive_pool2d_bwd": [_I, _I, _I, _I, _I, _I, _I, _P, _T, _P],
    "ea_gather_sum": [_I, _I, _I, _P, _P, _P, _P],
    "ea_table_bias_fwd": [_I, _I, _I, _I, _I, _F, _P, _P, _P, _P],
    "ea_multi_cast": [_I, _I, _P, _P, _P, _P],
    "ea_linear_w192_prepare": [_I, _P, _P, _P, _P, _P, _P, _P],
    "ea_linear_wsw": [_I, _I, _I, _I, _I, _I, _P, _I, _L, _P, _P, _P, _L, _P, _P, _P, _P],
    "ea_table_bias_bwd": [_I, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P],
    "ea_linear_supported": [_I, _I],
    "ea_linear": [_I, _I, _I, _I, _P, _I, _L, _P, _P, _P, _I, _L, _P, _P],
    "ea_linear_w32": [_I, _I, _I, _I, _P, _I, _L, _P, _I, _P, _P, _I, _L, _P, _P],
    "ea_linear_pool_supported": [_I] * 6,
    "ea_f32_attn_fwd": [_FA, _T, _T, _T, _T, _T, _P, _P, _P, _P, _P, _P, _T, _P, _P, _P],
    "ea_f32_attn_bwd": [_FA, _T, _T, _T, _T, _T, _P, _P, _P, _P, _P, _P, _T, _T, _P, _P, _T, _P, _P, _P, _P, _P, _P],
    "ea_f32_gather_mean_fwd": [_I, _I, _I, _I, _I, _I, _T, _P, _P, _P, _P],
    "ea_f32_gather_mean_bwd": [_I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "ea_linear_dgrad_supported": [_I, _I],
    "ea_linear_dgrad": [_I, _I, _I, _I, _P, _L, _P, _I, _P, _I, _L, _P],
    "ea_linear_dgrad_finish": [_I, _I, _I, _I, _I, _I, _F, _P, _L, _P, _L, _P, _I, _P, _I, _L, _P, _P, _P, _P, _P, _P],
    "ea_linear_w32_pool": [_I] * 7 + [_P, _I, _L, _P, _P, _P, _L, _P, _P, _P, _P, _P],
    "ea_wgrad_parts": [_I, _I, _I],
    "ea_wgrad": [_I, _I, _I, _I, _P, _P, _P, _P, _L, _P],
    "ea_wgrad_pair_parts": [_I, _I, _I, _I, _I],
    "ea_layernorm_parts": [_I],
    "ea_layernorm_fwd": [_I, _I, _I, _P, _P, _P, _F, _P, _P, _P],
    "ea_layernorm_bwd": [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "ea_wgrad_pair": [_I, _I, _I, _I, _P, _P, _P, _P, _L, _I, _I, _P, _P, _P, _P, _L, _P],
    "ea_part_sum": [_I, _I, _L, _P, _P, _P],
    "ea_multi_sum": [_I, _P, _P, _P, _P, _P, _P],
    "ea_lara_layer_ws": [_LL, _I],
    "ea_lara_layer_fwd": [_LL, _T, _T, _T, _P, _P, _P, _T, _P, _P, _I, _P],
    "ea_lara_layer_bwd": [_LL, _T, _T, _T, _P, _P, _P, _T, _T, _T, _T, _P, _P, _P, _P],
    "ea_lara_layer_bwd2": [_LL, _T, _T, _T, _P, _P, _P, _T, _T, _T, _T, _P, _P, _P, _I, _P],
    "ea_eva_layer_ws": [_EL, _I],
    "ea_eva_layer_fwd": [_EL, _T, _T, _T, _P, _P, _P, _T, _P, _I, _P],
    "ea_eva_layer_bwd": [_EL, _T, _T, _T, _P, _P, _P, _T, _T, _T, _T, _T, _P, _P, _P, _P, _P],
    "ea_eva_layer_bwd2": [_EL, _T, _T, _T, _P, _P, _P, _T, _T, _T, _T, _T, _P, _P, _P, _P, _I, _P],
    "ea_scatter_parts": [_SG],
    "ea_scatter_kmax": [_SG, _T, _P, _P, _P, _P],
    "ea_scatter_kv": [_SG, _T, _T, _P, _P, _P, _P, _P, _P],
    "ea_scatter_fwd": [_SG, _T, _T, _T, _P, _P, _P, _P, _P, _T, _P, _T, _P, _P],
    "ea_scatter_bwd_parts": [_SG],
    "ea_scatter_bwd_window": [_SG, _T, _T, _T, _P, _P, _P, _P, _P, _T, _P, _P, _T, _T, _T, _T, _T, _P, _P, _P, _P],
    "ea_scatter_bwd_global": [_SG, _T, _T, _P, _P, _P, _P, _P, _T, _T, _P],
    "ea_performer_parts": [_PG],
    "ea_performer_kmax": [_PG, _T, _P, _P, _P],
    "ea_performer_kv": [_PG, _T, _T, _P, _P, _P, _P, _P, _P],
    "ea_performer_out": [_PG, _T, _P, _P, _P, _T, _P],
    "ea_performer_bwd_q": [_PG, _T, _T, _T, _P, _P, _P, _T, _P, _P, _P, _P],
    "ea_performer_bwd_qstats": [_PG, _T, _T, _P, _P, _P, _P, _P, _P, _P],
    "ea_performer_bwd_k": [_PG, _T, _T, _P, _P, _P, _P, _P, _T, _T, _P],
    "ea_lara_seglin_groups": [_G],
    "ea_lara_seglin_fwd": [_G, _T, _T] + [_P] * 11,
    "ea_lara_seglin_bwd": [_G, _T, _T] + [_P] * 10 + [_T, _T, _P, _P, _P, _P],
    "ea_lara_seglin_bwd_fin": [_G, _T, _T] + [_P] * 10 + [_T, _T, _P, _P, _P, _P, _P, _P, ctypes.c_int32, ctypes.c_float, _P],
    "ea_lara_fold_fwd": [_I, _I, _I] + [_P] * 11,
    "ea_lara_fold_parts": [_I],
    "ea_lara_fold_bwd": [_I, _I, _P, _P, _P, _P, _P, _L] + [_P] * 10,
    "ea_performer_f32_parts": [_PG],
    "ea_performer_f32_kmax": [_PG, _T, _P, _P, _P],
    "ea_performer_f32_kv": [_PG, _T, _T, _P, _P, _P, _P, _P, _P],
    "ea_performer_f32_out": [_PG, _T, _P, _P, _P, _T, _P],
    "ea_performer_f32_bwd_q": [_PG, _T, _T, _P, _P, _P, _T, _P, _P, _P],
    "ea_performer_f32_bwd_k": [_PG, _T, _T, _P, _P, _P, _P, _P, _T, _T, _P],
    "ea_softmax_attn_fwd": [_I, _I, _I, _I, _I, _F, _T, _T, _T, _P, _T, _P, _P, _F, _I, _P],
    "ea_softmax_sample": [_I, _I, _I, _I, _I, _F, _T, _T, _P, _P, _P],
    "ea_softmax_attn_bwd": [_I, _I, _I, _I, _I, _F, _T, _T, _T, _P, _T, _T, _P, _P, _T, _T, _T, _P, _F, _I, _P],
    "ea_window_bias_ld": [_G],
    "ea_window_bwd_parts": [_G],
    "ea_window_bwd_needs_bias_t": [_G],
    "ea_window_bwd_acc_slices": [_G],
    "ea_window_bwd_bias_parts": [_G],
    "ea_window_bwd_query_blocks": [_G],
}

_lib = None


def lib():
    """Load libea_hip.so once; raise (never fall back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                "efficient_attention (MI355X build): %s not found. Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` at the repo root; there is "
                "no CPU fallback for the attention cores." % _LIB_PATH)
        cdll = ctypes.CDLL(_LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(cdll, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        cdll.ea_lara_landmarks_saved_floats.restype = ctypes.c_int64
        cdll.ea_lara_layer_ws.restype = ctypes.c_int64
        cdll.ea_eva_layer_ws.restype = ctypes.c_int64
        cdll.ea_version.restype = ctypes.c_char_p
        cdll.ea_abi_version.restype = ctypes.c_int32
        if cdll.ea_abi_version() != ABI_VERSION:
            raise RuntimeError("%s has ABI version %d, this package binds version %d -- rebuild it "
                               "(__graft_entry__.build())" % (_LIB_PATH, cdll.ea_abi_version(), ABI_VERSION))
        _lib = cdll
    return _lib


def version():
    return lib().ea_version().decode()


def _check(rc, what):
    if rc != 0:
        kind = {-1: "bad argument", -2: "unsupported geometry"}.get(rc, "hipError_t %d" % rc)
        raise RuntimeError("%s failed: %s" % (what, kind))


def require_cuda(t, what):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise RuntimeError(
            "efficient_attention (MI355X build): %s must be a CUDA/HIP tensor -- the attention "
            "cores are HIP kernels and have no CPU fallback." % what)


def io_dtype(t):
    if t.dtype not in _DTYPES:
        raise RuntimeError("attention cores take bf16 or fp16 tensors, got %s" % t.dtype)
    return _DTYPES[t.dtype]


def t4(t):
    """[B, H, N, D] view (any strides, D contiguous) -> ea_t4."""
    assert t.dim() == 4 and t.stride(3) == 1, (t.shape, t.stride())
    return ea_t4(t.data_ptr(), t.stride(0), t.stride(1), t.stride(2))


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """hipStream_t of torch's current stream on the current device (the launches go where torch's own would)."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def make_geom(B, H, N, D, dtype, attn_2d, seq_shape, window, ext, chunk=0, L=0, causal=0, lm_base=0):
    gh, gw = (seq_shape if attn_2d else (1, N))
    return ea_geom(B, H, N, D, dtype, 1 if attn_2d else 0, gh, gw, window, ext, chunk, L,
                   float(D) ** -0.5, causal, lm_base)


class KernelTimer:
    """Optional HIP-event bracket around every C-ABI launch (used by bench.py's instrumented
    pass and tools/): events are recorded on the stream the kernel is launched on."""

    def __init__(self):
        self.enabled = False
        self.records = []

    def enable(self):
        self.enabled, self.records = True, []

    def disable(self):
        self.enabled = False

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, a, b in self.records:
            ms = a.elapsed_time(b)
            st = out.setdefault(name, {"n": 0, "total_ms": 0.0})
            st["n"] += 1
            st["total_ms"] += ms
        for st in out.values():
            st["avg_ms"] = st["total_ms"] / st["n"]
        return out


KERNEL_TIMER = KernelTimer()


_FN = {}


def call_as(label, name, *args):
    """call(name, ...) timed under `label` (one C-ABI entry, several shapes: the label tells them apart)."""
    if not KERNEL_TIMER.enabled:
        return call(name, *args)
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    rc = getattr(lib(), name)(*args)
    b.record()
    KERNEL_TIMER.records.append((label, a, b))
    _check(rc, name)


def call(name, *args):
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(lib(), name)
    if not KERNEL_TIMER.enabled:
        rc = fn(*args)
        if rc != 0:
            _check(rc, name)
        return
    if KERNEL_TIMER.enabled:
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        rc = getattr(lib(), name)(*args)
        b.record()
        KERNEL_TIMER.records.append((name, a, b))
        _check(rc, name)
        return
    _check(getattr(lib(), name)(*args), name)


_QUERY_CACHE = {}


def query(name, geom):
    """Host-side plan query (a pure function of the geometry struct): memoised -- an eager step asks the same five questions
    about the same geometry every backward."""
    key = (name, bytes(geom))
    v = _QUERY_CACHE.get(key)
    if v is None:
        v = getattr(lib(), name)(ctypes.byref(geom))
        if v < 0:
            _check(v, name)
        if len(_QUERY_CACHE) < 4096:
            _QUERY_CACHE[key] = v
    return v
