"""-m gpu: the N > 1 step with the HIP path IN it (the CPU gloo test stands the oracle in for the
forward).  The GPU box has one MI355X, so the two ranks are two processes sharing cuda:0 and the collective runs on gloo
(RCCL refuses two ranks on one device); everything else is the bench's data-parallel step: the product module's HIP forward /
backward under bf16 autocast on each rank's batch shard, FlatGradBucket.pack -> all_reduce -> sgd_step
(efficient_attention/data_parallel.py).  Checked on rank 0: the bucket's averaged gradients == the same module's gradients on
the full batch / world (the attention path shards over the batch with no data-path collective), and the
parameters after the step == before - lr * that average on both ranks."""
import os
import socket
import sys
import warnings

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

CASES = {
    "lara": dict(dim=192, num_heads=3, num_landmarks=49, proposal_gen="pool-mixed", mis_type="mis-opt", alpha_coeff=2.0),
    "eva": dict(dim=192, num_heads=3, window_size=7, attn_2d=True, use_rpe=True, num_landmarks=49, adaptive_proj="default"),
}


def _build(attn):
    import torch
    import efficient_attention as ea
    torch.manual_seed(7)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = ea.AttentionFactory.build_attention(attn, dict(CASES[attn]))
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.05 * torch.randn_like(p))
    return m.cuda().eval()                      # eval: no sampling noise, so a shard's rows equal the full batch's rows


def _worker(rank, world, port, attn, ret):
    for p in (ROOT, os.path.join(ROOT, "efficient-attention_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from efficient_attention import _native as nv
        from efficient_attention.data_parallel import FlatGradBucket
        torch.cuda.set_device(0)
        model = _build(attn)
        if rank != 0:                            # the broadcast must repair a diverged replica
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(1.0)
        bucket = FlatGradBucket(model.parameters())
        bucket.broadcast_parameters(0)
        gen = torch.Generator(device="cuda").manual_seed(123)
        x = torch.randn(8, 28, 28, 192, device="cuda", generator=gen)
        g = torch.randn(8, 28, 28, 192, device="cuda", generator=gen)
        shard = slice(rank * 4, rank * 4 + 4)
        calls = []
        real = nv.call
        nv.call = lambda nm, *a: (calls.append(nm), real(nm, *a))[1]
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = model(x[shard])
            (y.float() * g[shard]).sum().backward()
        finally:
            nv.call = real
        assert any(c.startswith("ea_" + attn) or c.startswith("ea_window") for c in calls), sorted(set(calls))   # the HIP cores ran
        before = [p.detach().clone() for p in model.parameters()]
        bucket.pack()
        bucket.all_reduce()
        avg = [v.clone() for v in bucket.averaged_grads()]
        bucket.sgd_step(0.5)
        torch.cuda.synchronize()
        for p, b0, a in zip(model.parameters(), before, avg):
            assert torch.allclose(p, b0 - 0.5 * a, rtol=1e-6, atol=1e-7)
        if rank == 0:
            ref = _build(attn)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                yr = ref(x)
            (yr.float() * g).sum().backward()
            worst = 0.0
            for (k, p), a in zip(ref.named_parameters(), avg):
                assert p.grad is not None, k
                full = p.grad / world
                err = (a - full).abs().max().item() / max(full.abs().max().item(), 1e-12)
                worst = max(worst, err)
            ret["worst"] = worst
            ret["hip_calls"] = len(calls)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("attn", list(CASES))
def test_two_ranks_hip_forward_backward_flat_bucket(attn):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(2, port, attn, ret), nprocs=2, join=True)
        # per-sample work is batch-independent and every reduction over the batch is an fp32 sum: the two halves differ
        # from the full batch by summation order only
        assert ret["hip_calls"] > 0 and ret["worst"] < 2e-3, dict(ret)
