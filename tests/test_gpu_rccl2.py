"""-m gpu, needs TWO devices (skipped on the 1-GPU lease): the bench's N > 1 step as the driver's scaling run executes it --
two ranks, one MI355X each, RCCL (`backend="nccl"`) over xGMI, the HIP forward / backward captured in hipGraphs, the flat
bucket's all-reduce as the one eager call, and the schedule (pipelined / three_part) picked by `select_schedule` with the
timings max-reduced over RCCL (until now that combination had only run with fakes).
Checked: both ranks pick the same schedule; after K steps of the selected schedule + `flush_schedule` the parameters on both
ranks are bit-identical to each other and equal (to fp32 summation order) to K plain steps of the same module on the
concatenated batch with the full-batch gradient / world -- the data-parallel contract of SURVEY.md 8(e)."""
import os
import socket
import sys
import warnings

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LR = 0.05
STEPS = 3

CASES = {
    "lara": dict(dim=192, num_heads=3, num_landmarks=49, proposal_gen="pool-mixed", mis_type="mis-opt", alpha_coeff=2.0),
    "eva": dict(dim=192, num_heads=3, window_size=7, attn_2d=True, use_rpe=True, num_landmarks=49, adaptive_proj="default"),
}


def _build(attn, dev):
    import torch
    import efficient_attention as ea
    torch.manual_seed(11)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = ea.AttentionFactory.build_attention(attn, dict(CASES[attn]))
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.05 * torch.randn_like(p))
    return m.to(dev).eval()                     # eval: no sampling noise, a shard's rows equal the full batch's rows


def _worker(rank, world, port, attn, ret):
    for p in (ROOT, os.path.join(ROOT, "efficient-attention_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from efficient_attention.data_parallel import FlatGradBucket, ddp_schedules, select_schedule, flush_schedule
        model = _build(attn, dev)
        if rank != 0:
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(1.0)
        params = list(model.parameters())
        bucket = FlatGradBucket(params, dev)
        bucket.broadcast_parameters(0)
        start = [p.detach().clone() for p in params]
        gen = torch.Generator(device=dev).manual_seed(123)
        x = torch.randn(8, 28, 28, 192, device=dev, generator=gen)
        g = torch.randn(8, 28, 28, 192, device=dev, generator=gen).to(torch.bfloat16)
        xs, gs = x[rank * 4:rank * 4 + 4].contiguous(), g[rank * 4:rank * 4 + 4].contiguous()

        def fwd_bwd():
            for p in params:
                p.grad = None
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = model(xs)
            y.backward(gs)

        def capture(fn):
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                fn()
            torch.cuda.current_stream().wait_stream(s)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                fn()
            return gr

        def prepare(fns):                        # bench.py's: everything but the collective is a hipGraph replay
            out = [f if f.__name__ == "reduce" else capture(f).replay for f in fns]
            torch.cuda.synchronize()
            return out, True

        def reduce_max(sec):
            t = torch.tensor([sec], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        # LR = 0 while the race runs: select_schedule times REAL steps (documented benchmark-only side effects); the
        # parameters must still be the broadcast ones when the checked steps start
        schemes0 = ddp_schedules(fwd_bwd, bucket, 0.0)
        name, _, captured, seen = select_schedule(schemes0, prepare, torch.cuda.synchronize, reduce_max, barrier=dist.barrier,
                                                  warm=2, timed=3)
        assert captured and set(seen) == {"pipelined", "three_part"}
        names = [None, None]
        dist.all_gather_object(names, name)
        assert names[0] == names[1], names
        for p, s0 in zip(params, start):
            assert torch.equal(p, s0)
        bucket.flat.zero_()
        run, _ = prepare(ddp_schedules(fwd_bwd, bucket, LR)[name])
        # (prepare's capture warm-up ran each part once eagerly and once in capture -- with a zero bucket `pipelined`'s update is
        #  a no-op, `three_part`'s apply used a zero or freshly reduced bucket: restore and start clean)
        with torch.no_grad():
            for p, s0 in zip(params, start):
                p.copy_(s0)
        bucket.flat.zero_()
        for _ in range(STEPS):
            for f in run:
                f()
        flush_schedule(name, bucket, LR)
        torch.cuda.synchronize()
        mine = torch.cat([p.detach().reshape(-1) for p in params])
        both = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        assert torch.equal(both[0], both[1])     # replicas stay bit-identical
        if rank == 0:
            ref = _build(attn, dev)
            worst = 0.0
            for _ in range(STEPS):
                for p in ref.parameters():
                    p.grad = None
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    yr = ref(x)
                yr.backward(g)
                with torch.no_grad():
                    for p in ref.parameters():
                        p.add_(p.grad, alpha=-LR / world)
            for (k, pr), p, s0 in zip(ref.named_parameters(), params, start):
                moved = (pr - s0).abs().max().item()
                err = (p - pr).abs().max().item() / max(moved, 1e-12)
                worst = max(worst, err)
            ret["worst"] = worst
            ret["schedule"] = name
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("attn", list(CASES))
def test_two_devices_rccl_hipgraph_schedule_selection(attn):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two devices (RCCL refuses two ranks on one)")
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(2, port, attn, ret), nprocs=2, join=True)
        # K steps on two half batches vs K steps on the full batch: per-sample work is batch-independent, the two differ by
        # fp32 summation order (and, from step 2 on, by what that does to bf16-rounded activations): relative to the distance
        # the parameters moved
        assert ret["worst"] < 2e-2, dict(ret)
