"""dev: capture the causal_eva LM layer fwd(+bwd) in a hipGraph and replay it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-attention_amd")]
import torch
import bench

B = int(sys.argv[1]); mode = sys.argv[2]; dim = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
if len(sys.argv) > 4 and int(sys.argv[4]):
    bench.OVERRIDES["window_size"] = int(sys.argv[4])
for kv in sys.argv[5:]:
    k, v = kv.split("=")
    bench.LM_ATTN_ARGS[k] = {"True": True, "False": False}.get(v, v if not v.isdigit() else int(v))
layer = bench.build_layer("causal_eva", dim, 8, (512,), "cuda")
layer.train(mode != "evalfwd")
x = torch.randn(512, B, dim, device="cuda", requires_grad=True)
g = torch.randn(512, B, dim, device="cuda").to(torch.bfloat16)
def fn():
    for p in layer.parameters():
        p.grad = None
    x.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = layer(x, x, x)[0]
    if mode == "full":
        y.backward(g)
for _ in range(2):
    fn()
torch.cuda.synchronize(); print("eager ok", flush=True)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    fn()
torch.cuda.current_stream().wait_stream(s)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    fn()
torch.cuda.synchronize(); print("captured", flush=True)
for i in range(3):
    gr.replay(); torch.cuda.synchronize(); print("replay ok", i, flush=True)
for i in range(10):
    gr.replay()
torch.cuda.synchronize(); print("10 back-to-back replays ok", flush=True)
