"""dev: run the causal_eva LM layer eagerly with a device sync after every C-ABI call."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-attention_amd")]
import torch
import bench
from efficient_attention import _native as nv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 18
train = (sys.argv[2] != "eval") if len(sys.argv) > 2 else True
real = nv.call
def call(name, *a):
    real(name, *a)
    torch.cuda.synchronize()
    print("ok", name, flush=True)
nv.call = call
import efficient_attention._ops as ops
ops.nv.call = call
layer = bench.build_layer("causal_eva", 1024, 8, (512,), "cuda")
layer.train(train)
x = torch.randn(512, B, 1024, device="cuda", requires_grad=True)
for it in range(3):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = layer(x, x, x)[0]
    torch.cuda.synchronize(); print("fwd done", it, flush=True)
    y.backward(torch.randn_like(y))
    torch.cuda.synchronize(); print("bwd done", it, flush=True)
