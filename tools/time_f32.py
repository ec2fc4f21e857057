"""Layer fwd+bwd in fp32 OUTSIDE autocast (the fp32-faithful cores, efficient_attention/_f32.py) next to the bf16-autocast step:
   python tools/time_f32.py   (GPU).  Prints ms per step and tokens/s for BASELINE.json configs[0] (EVA, x = [2,14,14,512], h = 8)
and the cfg3 geometry at batch 32."""
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-attention_amd")]
import torch  # noqa: E402
import efficient_attention as ea  # noqa: E402

warnings.simplefilter("ignore")
CASES = [("eva", (2, 14, 14, 512), dict(dim=512, num_heads=8, window_size=7, attn_2d=True, use_rpe=True, num_landmarks=49)),
         ("eva", (32, 28, 28, 192), dict(dim=192, num_heads=3, window_size=7, attn_2d=True, use_rpe=True, num_landmarks=49)),
         ("lara", (32, 28, 28, 192), dict(dim=192, num_heads=3, num_landmarks=49, proposal_gen="pool-mixed", mis_type="mis-opt", alpha_coeff=2.0)),
         ("local", (32, 28, 28, 192), dict(dim=192, num_heads=3, window_size=7, attn_2d=True, use_rpe=True)),
         ("softmax", (32, 28, 28, 192), dict(dim=192, num_heads=3))]
for attn, shape, args in CASES:
    m = ea.AttentionFactory.build_attention(attn, args).cuda().train()
    x = torch.randn(*shape, device="cuda", requires_grad=True)
    g = torch.randn(*shape, device="cuda")
    for mode in ("fp32", "bf16"):
        def step():
            for p in m.parameters():
                p.grad = None
            if mode == "bf16":
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    y = m(x)
            else:
                y = m(x)
            y.backward(g.to(y.dtype))
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        ntok = shape[0] * shape[1] * shape[2]
        print("%-8s %-18s %s  %.3f ms/step  %.2f M tokens/s" % (attn, shape, mode, ms, ntok / ms / 1e3))
