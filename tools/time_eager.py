"""dev tool: eager (un-captured) step time of the bench layer, composite C-ABI entry on / off."""
import sys, os, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-attention_amd")]
import torch
import bench
warnings.simplefilter("ignore")
attn = sys.argv[1] if len(sys.argv) > 1 else "lara"
m = bench.build_layer(attn, 192, 3, (28, 28), "cuda"); m.train()
x = torch.randn(128, 28, 28, 192, device="cuda", requires_grad=True)
g = torch.randn(128, 28, 28, 192, device="cuda").bfloat16()
def step():
    for p in m.parameters(): p.grad = None
    x.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = m(x)
    y.backward(g)
for mode in ("1", "0", "1", "0"):
    os.environ["EA_LARA_COMPOSITE"] = mode
    for _ in range(10): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): step()
    torch.cuda.synchronize()
    print("composite", mode, "eager ms/step %.4f" % ((time.perf_counter() - t0) / 200 * 1e3))
    # host-only cost: time to ISSUE 200 steps (no sync inside)
import cProfile, pstats
os.environ["EA_LARA_COMPOSITE"] = "1"
pr = cProfile.Profile()
pr.enable()
for _ in range(100): step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime"); st.print_stats(22)
