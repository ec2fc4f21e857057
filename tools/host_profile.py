"""dev tool (GPU): cProfile of the eager layer step -- where the HOST time of an un-captured step goes.
  python tools/host_profile.py eva [cfg3]"""
import cProfile
import os
import pstats
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench

attn = sys.argv[1] if len(sys.argv) > 1 else "eva"
B, C, H, seq = 128, 192, 3, (28, 28)
dev = torch.device("cuda:0")
layer = bench.build_layer(attn, C, H, seq, dev)
layer.train()
x = torch.randn(*((B,) + tuple(seq) + (C,)), device=dev, requires_grad=True)
g = torch.randn(*((B,) + tuple(seq) + (C,)), device=dev).to(torch.bfloat16)
params = list(layer.parameters())


def step():
    for prm in params:
        prm.grad = None
    x.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = layer(x)
    y.backward(g)


for _ in range(10):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
