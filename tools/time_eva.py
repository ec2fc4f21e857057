"""Quick per-kernel timing of the EVA core at the bench shapes (dev tool, GPU only)."""
import sys, os, ctypes, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-attention_amd")]
import torch
import efficient_attention as ea
from efficient_attention import _ops, _native as nv

def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3   # us

def run(B, H, Wg, w, L, d=64):
    N = Wg * Wg
    dev = "cuda"
    torch.manual_seed(0)
    qkv5 = (torch.randn(B, N, 3, H, d, device=dev) * 0.5).to(torch.bfloat16)
    r = int((N // L) ** 0.5)
    Lc = (Wg // r) ** 2
    geom = nv.make_geom(B, H, N, d, 0, True, (Wg, Wg), w, 0, r, Lc)
    q, k, v = _ops._qkv_views(qkv5)
    tq, tk, tv = nv.t4(q), nv.t4(k), nv.t4(v)
    qmean = torch.empty(B, H, Lc, d, device=dev); kmean = torch.empty_like(qmean)
    omega = torch.randn(B, H, Lc, d, device=dev); beta = torch.empty_like(omega)
    rfk = torch.randn(B, H, Lc, d, device=dev)
    bias = torch.randn(H, w * w, w * w, device=dev) * 0.5
    bias_p = _ops._bias_padded(bias, geom)
    out, lse = _ops._window_fwd(geom, qkv5, rfk, beta.normal_(), bias_p, None)
    dout = torch.randn_like(out); dqkv5 = torch.empty_like(qkv5)
    dq, dk, dv = _ops._qkv_views(dqkv5); tdq, tdk, tdv = nv.t4(dq), nv.t4(dk), nv.t4(dv)
    dbeta = torch.randn_like(beta); dom = torch.empty_like(beta)
    st = nv.stream()
    res = {}
    res["mean_fwd"] = timeit(lambda: nv.call("ea_eva_chunk_mean_fwd", ctypes.byref(geom), ctypes.byref(tq), ctypes.byref(tk), None, nv.ptr(qmean), nv.ptr(kmean), st))
    res["beta_fwd"] = timeit(lambda: nv.call("ea_eva_beta_fwd", ctypes.byref(geom), ctypes.byref(tk), ctypes.byref(tv), None, nv.ptr(omega), nv.ptr(beta), st))
    res["win_fwd"] = timeit(lambda: _ops._window_fwd(geom, qkv5, rfk, beta, bias_p, None))
    res["win_fwd_nobias"] = timeit(lambda: _ops._window_fwd(geom, qkv5, rfk, beta, None, None))
    res["win_bwd"] = timeit(lambda: _ops._window_bwd(geom, qkv5, rfk, beta, bias_p, None, out, dout, lse, dqkv5))
    res["win_bwd_nobias"] = timeit(lambda: _ops._window_bwd(geom, qkv5, rfk, beta, None, None, out, dout, lse, dqkv5))
    res["beta_bwd"] = timeit(lambda: nv.call("ea_eva_beta_bwd", ctypes.byref(geom), ctypes.byref(tk), ctypes.byref(tv), None, nv.ptr(omega), nv.ptr(beta), nv.ptr(dbeta), ctypes.byref(tdk), ctypes.byref(tdv), nv.ptr(dom), st))
    res["mean_bwd"] = timeit(lambda: nv.call("ea_eva_chunk_mean_bwd", ctypes.byref(geom), nv.ptr(qmean), nv.ptr(kmean), None, ctypes.byref(tdq), ctypes.byref(tdk), st))
    unit = B * H * N * d * 2  # bytes of one [B,H,N,d] bf16 tensor
    print("B=%d H=%d N=%d w=%d L=%d  unit=%.1f MB" % (B, H, N, w, Lc, unit / 1e6))
    for kname, us in res.items():
        print("  %-16s %8.1f us" % (kname, us))
    print("  fwd 4u @ win_fwd: %.2f TB/s ; bwd 8u @ win_bwd: %.2f TB/s" % (4 * unit / res["win_fwd"] / 1e6, 8 * unit / res["win_bwd"] / 1e6))
    # module-level fwd+bwd
    m = ea.AttentionFactory.build_attention("eva", dict(dim=H * d, num_heads=H, window_size=w, attn_2d=True, use_rpe=True, num_landmarks=L)).cuda()
    x = torch.randn(B, Wg, Wg, H * d, device=dev, requires_grad=True)
    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(x)
        y.float().sum().backward()
    us = timeit(step, n=10)
    print("  module fwd+bwd %.1f us -> %.2f M tok/s" % (us, B * N / us))

import warnings; warnings.simplefilter("ignore")
run(128, 3, 14, 7, 49)
run(128, 3, 28, 7, 49)
run(1024, 3, 14, 7, 49)
