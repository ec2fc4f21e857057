import sys, os, time, warnings
sys.path[:0] = ['/root/repo', '/root/repo/efficient-attention_amd']
import torch, cProfile, pstats
import bench
warnings.simplefilter("ignore")
m = bench.build_layer("eva", 192, 3, 28, "cuda"); m.train()
x = torch.randn(128, 28, 28, 192, device="cuda", requires_grad=True)
g = torch.randn(128, 28, 28, 192, device="cuda").bfloat16()
def step():
    for p in m.parameters(): p.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = m(x)
    y.backward(g)
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize()
print("eager ms/step", (time.perf_counter() - t0) / 50 * 1e3, "EA_DGRAD_FIN", os.environ.get("EA_DGRAD_FIN"))
pr = cProfile.Profile(); pr.enable()
for _ in range(30): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
