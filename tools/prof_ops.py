"""dev tool (GPU): aten-op level profile of one layer step (which torch glue ops run around the HIP kernels).
  python tools/prof_ops.py lara cfg5"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from torch.profiler import profile, ProfilerActivity

attn = sys.argv[1] if len(sys.argv) > 1 else "lara"
wl = sys.argv[2] if len(sys.argv) > 2 else "cfg5"
B, C, H, seq = {"cfg5": (16, 512, 8, (4096,)), "cfg3": (32, 192, 3, (56, 56)), "cfg2": (128, 192, 3, (14, 14))}[wl]
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


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=60))
