"""dev tool (GPU): ea_linear against the library GEMM (+ the cast it folds in) at the bench shapes; checks the result too."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-attention_amd")]
import torch
import torch.nn.functional as F
from efficient_attention import _ops

# rows, in, out, a fp32?, y fp32?
shapes = [(100352, 192, 576, 1, 0), (100352, 192, 192, 0, 0), (100352, 192, 192, 0, 1), (25088, 192, 576, 1, 0),
          (100001, 64, 128, 1, 0), (50017, 128, 384, 0, 0), (30001, 256, 256, 1, 1), (77, 192, 576, 1, 0)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
NB = 3


def timeit(fn, n=20):
    for i in range(3):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for rows, K, NO, af32, yf32 in shapes:
    adt = torch.float32 if af32 else torch.bfloat16
    ydt = torch.float32 if yf32 else torch.bfloat16
    A = [torch.randn(rows, K, device="cuda").to(adt) for _ in range(NB)]
    w = (torch.randn(NO, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.randn(NO, device="cuda")
    y, ac = _ops.ea_linear(A[0], w, b, ydt, want_cast=bool(af32))
    a16 = A[0].bfloat16()
    if yf32:
        ref = torch.mm(a16, w.t(), out_dtype=torch.float32) + b.bfloat16().float()
    else:
        ref = F.linear(a16, w, b.bfloat16())
    err = (y.float() - ref.float()).abs().max().item()
    same = (y == ref).float().mean().item()
    cast_ok = bool(af32) and bool((ac == a16).all().item())
    t_ea = timeit(lambda i: _ops.ea_linear(A[i % NB], w, b, ydt, want_cast=bool(af32)))
    if yf32:
        t_lib = timeit(lambda i: torch.mm(A[i % NB].bfloat16() if af32 else A[i % NB], w.t(), out_dtype=torch.float32))
    else:
        bb = b.bfloat16()
        t_lib = timeit(lambda i: F.linear(A[i % NB].bfloat16() if af32 else A[i % NB], w, bb))
    mb = (rows * K * (4 if af32 else 2) + rows * NO * (4 if yf32 else 2) + (rows * K * 2 if af32 else 0)) / 1e6
    print("rows %d in %d out %d a_f32 %d y_f32 %d: ea %.1f us (%.0f GB/s) lib%s %.1f us | max err %.3g, identical %.4f, cast copy ok %s"
          % (rows, K, NO, af32, yf32, t_ea, mb / t_ea * 1e3, "+cast" if af32 else "", t_lib, err, same, cast_ok))
