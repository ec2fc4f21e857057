"""dev: ea_colsum_f32 at the shapes the LM / cfg5 steps call it with, and the two-stage view for tall matrices
([rows, cols] read as [rows / k, k * cols], then [k, cols]).  Run on the GPU box."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "efficient-attention_amd")]
from efficient_attention import _ops


def t(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def two_stage(x, k):
    rows, cols = x.shape
    p = _ops._colsum_raw(x.view(rows // k, k * cols))
    return _ops._colsum_raw(p.view(k, cols))


for rows, cols in [(18, 131072), (18, 196608), (72, 131072), (72, 32768), (144, 32768), (9216, 768), (65536, 768), (4096, 768),
                   (2304, 384), (1152, 6 * 64), (100352 // 64, 768)]:
    x = torch.randn(rows, cols, device="cuda")
    base = t(lambda: _ops._colsum_raw(x))
    line = "rows %6d cols %6d  %6.2f MB  one-stage %7.1f us" % (rows, cols, rows * cols * 4 / 1e6, base)
    ref = x.double().sum(0)
    for k in (4, 8, 12, 16, 32, 64):
        if rows % k == 0 and rows // k >= 16:
            us = t(lambda: two_stage(x, k))
            err = (two_stage(x, k).double() - ref).abs().max().item()
            line += " | k=%d %6.1f us (err %.1e)" % (k, us, err)
    print(line + " | auto %6.1f us" % t(lambda: _ops.colsum_f32(x)), flush=True)
