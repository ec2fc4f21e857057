"""One synthetic training step of a whole model, the way the reference's call sites run it
(vit/engine.py:40-69: autocast forward + loss, backward, optimizer step; main.sh:145-158,183: one
process per GPU, DistributedDataParallel).  Synthetic data of the named shape, random-init weights.

    wl = build_workload("model_cfg3", device)          # model, batch, loss, tokens per step
    run = make_step(wl, ddp=False, graph=False)         # callable: one optimizer step
"""
import contextlib

import torch
import torch.nn as nn
import torch.nn.functional as F

from .vision import deit_tiny, pvt_b2
from .sequence import wmt_en_de_encoder

# BASELINE.json configs 2-5 as whole models.  cfg4: 384^2 gives grids 96/48/24, which window 7 / 49
# landmarks do not divide (the reference asserts) -> window 8, 36 landmarks as recorded there.
WORKLOADS = {
    "model_cfg2": dict(kind="deit", patch=16, img=224, batch=128, attn="eva",
                       attn_args=dict(window_size=7, attn_2d=True, use_rpe=True, num_landmarks=49, adaptive_proj="default"),
                       desc="DeiT-tiny-p16 (12 blocks, N=196) EVA, ImageNet-224 batch 128"),
    "model_cfg3": dict(kind="deit", patch=8, img=224, batch=128, attn="lara",
                       attn_args=dict(num_landmarks=49, proposal_gen="pool-mixed", mis_type="mis-opt", alpha_coeff=2.0),
                       desc="DeiT-tiny-p8 (12 blocks, N=784) LARA mis-opt pool-mixed 49 landmarks, ImageNet-224 batch 128"),
    "model_cfg4": dict(kind="pvt", img=384, batch=32, attn="eva",
                       attn_args=dict(window_size=8, attn_2d=True, use_rpe=True, num_landmarks=36, adaptive_proj="default"),
                       desc="PvTv2-b2 (3-4-6-3 blocks) EVA w=8 L=36 + softmax last stage, 384x384 batch 32"),
    "model_cfg5": dict(kind="seq", seq=4096, batch=1, attn="lara",
                       attn_args=dict(num_landmarks=49, proposal_gen="adaptive-1d", mis_type="mis-opt"),
                       desc="transformer_wmt_en_de encoder (6 layers, 512/2048, 8 heads) LARA adaptive-1d, 4096 tokens"),
}


class Workload:
    def __init__(self, name, model, inputs, target, loss_fn, tokens, desc):
        self.name, self.model, self.inputs, self.target = name, model, inputs, target
        self.loss_fn, self.tokens, self.desc = loss_fn, tokens, desc


def attention_tokens(kind, img=None, patch=None, batch=1, seq=None):
    """Tokens that pass through attention layers in one step / number of attention layers."""
    if kind == "deit":
        return batch * (img // patch) ** 2
    if kind == "pvt":
        return batch * (img // 4) ** 2          # first-stage grid (the metric's N for the multi-scale model)
    return batch * seq


def build_workload(name, device, batch=None, attn=None, attn_args=None, num_classes=1000, seed=1234):
    cfg = dict(WORKLOADS[name])
    if batch:
        cfg["batch"] = batch
    if attn:
        cfg["attn"] = attn
        cfg["attn_args"] = attn_args or {}
    B = cfg["batch"]
    torch.manual_seed(seed)
    if cfg["kind"] == "deit":
        model = deit_tiny(cfg["attn"], cfg["attn_args"], cfg["patch"], img_size=cfg["img"], num_classes=num_classes)
        x = torch.randn(B, 3, cfg["img"], cfg["img"], device=device)
        y = torch.randint(0, num_classes, (B,), device=device)
        loss_fn = lambda m, inp, tgt: F.cross_entropy(m(inp).float(), tgt)      # noqa: E731
        tokens = attention_tokens("deit", cfg["img"], cfg["patch"], B)
    elif cfg["kind"] == "pvt":
        model = pvt_b2(cfg["attn"], cfg["attn_args"], img_size=cfg["img"], num_classes=num_classes)
        x = torch.randn(B, 3, cfg["img"], cfg["img"], device=device)
        y = torch.randint(0, num_classes, (B,), device=device)
        loss_fn = lambda m, inp, tgt: F.cross_entropy(m(inp).float(), tgt)      # noqa: E731
        tokens = attention_tokens("pvt", cfg["img"], batch=B)
    else:
        vocab = 32768
        model = wmt_en_de_encoder(cfg["attn"], cfg["attn_args"], vocab=vocab)
        x = torch.randint(2, vocab, (B, cfg["seq"]), device=device)
        y = torch.randint(2, vocab, (cfg["seq"], B), device=device)

        def loss_fn(m, inp, tgt):
            core = m.module if hasattr(m, "module") else m
            h = m(inp)
            return F.cross_entropy(core.logits(h).float().view(-1, vocab), tgt.view(-1))
        tokens = attention_tokens("seq", batch=B, seq=cfg["seq"])
    model = model.to(device)
    model.train()
    return Workload(name, model, x, y, loss_fn, tokens, cfg["desc"])


def make_step(wl, optimizer=None, ddp_model=None, dtype=torch.bfloat16, lr=1e-3):
    """-> step(): zero grads, autocast forward + loss, backward, optimizer step (vit/engine.py:47-64).
    `ddp_model`: the DistributedDataParallel wrapper of wl.model when data-parallel."""
    model = ddp_model if ddp_model is not None else wl.model
    opt = optimizer or torch.optim.SGD(wl.model.parameters(), lr=lr, momentum=0.9)
    dev_type = wl.inputs.device.type
    amp = torch.autocast(dev_type, dtype=dtype) if dev_type == "cuda" else contextlib.nullcontext()

    def step():
        opt.zero_grad(set_to_none=step.set_to_none)
        with amp:
            loss = wl.loss_fn(model, wl.inputs, wl.target)
        loss.backward()
        opt.step()
        return loss
    step.optimizer = opt
    step.set_to_none = True
    return step


def capture_step(step, warmup=3):
    """Capture `step` in a hipGraph (after `warmup` eager runs on a side stream); returns replay()."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(warmup):
            step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    step.set_to_none = False              # gradients keep their storage across replays: zeroed in place
    with torch.cuda.graph(g):
        step()
    return g.replay


def wrap_ddp(model, device, bucket_cap_mb=25):
    """DistributedDataParallel over the default (RCCL / gloo) process group: bucketed gradient all-reduce
    overlapped with backward, gradients as views of the buckets (no extra copy)."""
    ids = [device.index] if device.type == "cuda" else None
    return nn.parallel.DistributedDataParallel(model, device_ids=ids, bucket_cap_mb=bucket_cap_mb,
                                               gradient_as_bucket_view=True, broadcast_buffers=False)
