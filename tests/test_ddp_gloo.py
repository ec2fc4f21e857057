"""-m "not gpu": the N > 1 path on CPU -- two gloo ranks, one process per rank, rendezvous on
127.0.0.1.  The attention path shards over the batch with no data-path collective;
the only exchange is the parameter-gradient all-reduce of DistributedDataParallel.  The product
module supplies the parameters (every one must receive a gradient on every rank, as DDP
requires); its forward is stood in for by the oracle on CPU, since the HIP cores need a GPU.
Checked: DDP-averaged gradients of the two half-batches == gradients of the full batch."""
import os
import socket
import warnings

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle

CASES = {
    "eva": dict(dim=64, num_heads=2, window_size=7, attn_2d=True, use_rpe=True, num_landmarks=4),
    "lara": dict(dim=64, num_heads=2, num_landmarks=4, proposal_gen="pool-mixed", mis_type="mis-opt",
                 alpha_coeff=2.0),
}


class OracleBacked(torch.nn.Module):
    """Product module's parameters + the oracle's CPU forward (test stand-in for the HIP cores)."""

    def __init__(self, attn, args, inner):
        super().__init__()
        self.attn, self.args, self.inner = attn, args, inner

    def forward(self, x):
        params = dict(self.inner.named_parameters())
        params.update(dict(self.inner.named_buffers()))
        return oracle.module_forward(self.attn, self.args, params, x, None, training=False)


def _build(attn):
    import efficient_attention as ea
    torch.manual_seed(7)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        inner = ea.AttentionFactory.build_attention(attn, dict(CASES[attn]))
    with torch.no_grad():                       # make zero-initialised biases/tables matter
        for p in inner.parameters():
            p.add_(0.05 * torch.randn_like(p))
    return OracleBacked(attn, CASES[attn], inner)


def _worker(rank, world, port, attn, ret, flat=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        model = _build(attn)
        torch.manual_seed(123)
        x = torch.randn(4, 14, 14, 64)
        g = torch.randn(4, 14, 14, 64)
        shard = slice(rank * 2, rank * 2 + 2)                     # weak-scaling style batch shard
        if flat:
            # the bench's N > 1 path: one flat gradient bucket, one all-reduce, SGD from the bucket
            from efficient_attention.data_parallel import FlatGradBucket
            if rank != 0:                                         # broadcast must repair a diverged replica
                with torch.no_grad():
                    for p in model.parameters():
                        p.add_(1.0)
            bucket = FlatGradBucket(model.parameters())
            bucket.broadcast_parameters(0)
            before = [p.detach().clone() for p in model.parameters()]
            (model(x[shard]) * g[shard]).sum().backward()
            bucket.pack()
            bucket.all_reduce()
            grads = {k: v.clone() for (k, _), v in zip(model.named_parameters(), bucket.averaged_grads())}
            bucket.sgd_step(0.5)
            for p, b0, (k, _) in zip(model.parameters(), before, model.named_parameters()):
                assert torch.allclose(p, b0 - 0.5 * grads[k], rtol=1e-6, atol=1e-7), k
        else:
            ddp = torch.nn.parallel.DistributedDataParallel(model)
            (ddp(x[shard]) * g[shard]).sum().backward()
            grads = {k: p.grad.clone() for k, p in model.named_parameters()}
        if rank == 0:
            ref = _build(attn)
            (ref(x) * g).sum().backward()
            worst = 0.0
            for k, p in ref.named_parameters():
                assert p.grad is not None, k
                full = p.grad / world                             # DDP averages over ranks
                err = (grads[k] - full).abs().max().item() / max(full.abs().max().item(), 1e-12)
                worst = max(worst, err)
            ret["worst"] = worst
    finally:
        dist.destroy_process_group()


def _run(attn, flat):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(2, port, attn, ret, flat), nprocs=2, join=True)
        assert ret["worst"] < 1e-4, dict(ret)


@pytest.mark.parametrize("attn", list(CASES))
def test_two_rank_gloo_ddp_matches_full_batch(attn):
    _run(attn, False)


@pytest.mark.parametrize("attn", list(CASES))
def test_two_rank_gloo_flat_bucket_matches_full_batch(attn):
    """bench.py's N > 1 path (FlatGradBucket): parameter broadcast, one all-reduce of all gradients,
    averaged gradients == full-batch gradients, SGD step applied from the bucket."""
    _run(attn, True)


# ---- the schedule selection of bench.py's N > 1 path: factored into
# efficient_attention.data_parallel.select_schedule so that it runs here, without a second GPU ----
class _FakeClock:
    def __init__(self):
        self.t = 0.0

    def __call__(self):
        return self.t


def test_select_schedule_picks_the_fastest_and_runs_every_part_in_order():
    from efficient_attention.data_parallel import select_schedule
    clock, log = _FakeClock(), []

    def part(name, cost):
        def f():
            log.append(name)
            clock.t += cost
        f.__name__ = name
        return f
    schemes = {"pipelined": [part("update_and_pack", 2.0), part("reduce", 1.0)],
               "three_part": [part("pack", 1.5), part("reduce", 1.0), part("apply", 0.25)]}
    prepared = []

    def prepare(fns):
        prepared.append([f.__name__ for f in fns])
        return fns, True
    syncs = []
    name, run, captured, seen = select_schedule(schemes, prepare, lambda: syncs.append(1), lambda s: s, clock=clock, warm=3, timed=5)
    assert name == "three_part" and captured and [f.__name__ for f in run] == ["pack", "reduce", "apply"]
    assert seen == {"pipelined": 15.0, "three_part": 13.75}
    assert prepared == [["update_and_pack", "reduce"], ["pack", "reduce", "apply"]]
    assert log == ["update_and_pack", "reduce"] * 8 + ["pack", "reduce", "apply"] * 8       # 3 warm + 5 timed steps each
    assert len(syncs) == 4                                                                     # before and after each timed block
    # ties go to the first schedule; a forced schedule is not raced; an unknown one is an error
    tie = {"a": [part("x", 1.0)], "b": [part("y", 1.0)]}
    assert select_schedule(tie, prepare, lambda: None, lambda s: s, clock=clock)[0] == "a"
    log.clear()
    assert select_schedule(schemes, prepare, lambda: None, lambda s: s, clock=clock, forced="pipelined")[0] == "pipelined"
    assert "pack" not in log
    with pytest.raises(KeyError):
        select_schedule(schemes, prepare, lambda: None, lambda s: s, forced="nope")


def _sched_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from efficient_attention.data_parallel import FlatGradBucket, ddp_schedules, select_schedule
        torch.manual_seed(11)
        lin = torch.nn.Linear(8, 8)
        x = torch.randn(4, 8)[rank * 2: rank * 2 + 2]
        bucket = FlatGradBucket(list(lin.parameters()))
        bucket.broadcast_parameters(0)
        clock = _FakeClock()
        # rank-dependent cost model: rank 0 finds 'pipelined' faster, rank 1 finds it much slower -> the MAX over ranks must
        # make both ranks choose 'three_part'
        cost = {"update_and_pack": [1.0, 9.0][rank], "pack": [2.0, 2.0][rank], "apply": 0.5, "reduce": 0.1}

        def fwd_bwd():
            for p in lin.parameters():
                p.grad = None
            lin(x).sum().backward()
        schemes = ddp_schedules(fwd_bwd, bucket, 0.1)

        def prepare(fns):
            def wrap(f):
                def g():
                    f()
                    clock.t += cost[f.__name__]
                g.__name__ = f.__name__
                return g
            return [wrap(f) for f in fns], False

        def reduce_max(sec):
            t = torch.tensor([sec], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        name, run, captured, seen = select_schedule(schemes, prepare, lambda: None, reduce_max, barrier=dist.barrier, clock=clock)
        for f in run:                                            # the chosen schedule still steps
            f()
        ret[rank] = (name, seen, [float(p.detach().abs().sum()) for p in lin.parameters()])
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_schedule_selection_agrees_across_ranks():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_sched_worker, args=(2, port, ret), nprocs=2, join=True)
        r0, r1 = ret[0], ret[1]
    assert r0[0] == r1[0] == "three_part", (r0, r1)
    assert r0[1] == r1[1]                                          # identical (max-reduced) timings on both ranks
    assert r0[1]["pipelined"] == pytest.approx(5 * 9.1) and r0[1]["three_part"] == pytest.approx(5 * 2.6)
    assert r0[2] == pytest.approx(r1[2])                           # replicas stay in step through the selection runs


# ---- bench.py's clock pre-warm with more than one rank: the ranks leave it after the same number of steps ----
def _prewarm_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from efficient_attention.data_parallel import prewarm_replays
        clock = _FakeClock()
        buf = torch.zeros(4)
        count = [0]

        def run():                                               # a step = one collective; rank 1's clock runs 5x faster
            dist.all_reduce(buf)
            count[0] += 1
            clock.t += [1e-3, 5e-3][rank]

        def agree(done):
            flag = torch.tensor([1 if done else 0], dtype=torch.int32)
            dist.broadcast(flag, 0)
            return bool(flag.item())
        n = prewarm_replays(run, 50.0, lambda: None, agree=agree, clock=clock)
        dist.barrier()                                           # would hang / mismatch if the collective counts differed
        ret[rank] = (n, count[0])
    finally:
        dist.destroy_process_group()


def test_prewarm_replays_same_step_count_on_every_rank():
    from efficient_attention.data_parallel import prewarm_replays
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_prewarm_worker, args=(2, port, ret), nprocs=2, join=True)
        assert ret[0] == ret[1] == (56, 56), dict(ret)           # rank 0's clock: 50 ms / 1 ms per step, in chunks of 8
    # single rank: its own clock; nothing to replay when switched off
    clock, calls = _FakeClock(), []

    def run():
        calls.append(1)
        clock.t += 2e-3
    assert prewarm_replays(run, 50.0, lambda: None, clock=clock) == 32 and len(calls) == 32
    assert prewarm_replays(run, 0.0, lambda: None, clock=clock) == 0
