"""Data parallelism for the attention layers: one process per GPU, batch sharded, no data-path
collective (every (batch element, head) is independent -- SURVEY.md 8e).  The only exchange is the
parameter-gradient all-reduce, done here the MI355X way: ONE flat fp32 bucket per step (the layer
has ~0.16 M parameters, far below the size where xGMI ring bandwidth matters, so a single
latency-bound RCCL call beats per-bucket hooks), with the 1/world averaging folded into the
optimiser step.  Numerically this is DistributedDataParallel + SGD; unlike the DDP wrapper it adds
no per-iteration host work, so the forward/backward and the update can stay inside captured
hipGraphs with the all-reduce as the only eager call between them (bench.py)."""
import time

import torch
import torch.distributed as dist


class FlatGradBucket:
    def __init__(self, params, device=None):
        self.params = [p for p in params]
        device = device or self.params[0].device
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=torch.float32, device=device)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def broadcast_parameters(self, src=0):
        """Same initial parameters on every rank (what DDP does at construction)."""
        for p in self.params:
            dist.broadcast(p.data, src)

    def pack(self):
        """Copy every parameter's .grad into the bucket (one multi-tensor kernel)."""
        grads = []
        for p in self.params:
            if p.grad is None:
                raise RuntimeError("FlatGradBucket.pack: a parameter received no gradient")
            grads.append(p.grad)
        torch._foreach_copy_(self.views, grads)

    def all_reduce(self):
        dist.all_reduce(self.flat)             # SUM over ranks; averaged in sgd_step / averaged_grads

    def averaged_grads(self):
        world = dist.get_world_size()
        return [v / world for v in self.views]

    def sgd_step(self, lr):
        """p -= lr * mean-over-ranks(grad), straight from the bucket."""
        with torch.no_grad():
            torch._foreach_add_(self.params, self.views, alpha=-lr / dist.get_world_size())


def ddp_schedules(fwd_bwd, bucket, lr):
    """The two equivalent schedules of one data-parallel step around a FlatGradBucket (bench.py, N > 1):
      pipelined : [update from the previous step's bucket + forward/backward + pack] | all-reduce
      three_part: [forward/backward + pack] | all-reduce | [update]
    Each is a list of callables; the one named `reduce` is the eager collective, the others may be captured in hipGraphs.
    With the bucket zero before the first step both hold one update, one forward/backward and one all-reduce per step."""
    def update_and_pack():
        bucket.sgd_step(lr)
        fwd_bwd()
        bucket.pack()

    def pack():
        fwd_bwd()
        bucket.pack()

    def reduce():
        bucket.all_reduce()

    def apply():
        bucket.sgd_step(lr)
    return {"pipelined": [update_and_pack, reduce], "three_part": [pack, reduce, apply]}


def select_schedule(schemes, prepare, sync, reduce_max, barrier=None, clock=time.perf_counter, warm=3, timed=5, forced=None):
    """Pick the faster of several equivalent step schedules by MEASURING them: for each (name, callables) of `schemes`,
    `prepare(callables)` -> (runnable callables, captured?) (e.g. hipGraph capture of everything but the collective), `warm`
    untimed steps, [barrier], sync, `timed` steps between two `clock()` readings with a `sync()` before the second, and
    `reduce_max(seconds)` -> the maximum over all ranks -- every rank therefore sees the same figures and takes the same
    decision (ties: the first schedule).  `forced` names the schedule to take without timing the others.
    Returns (name, runnable callables, captured?, {name: seconds per `timed` steps}).
    BENCHMARK-ONLY side effects: the candidates are timed by running REAL steps on the one shared bucket, so
    an update may be dropped (a candidate's pack() overwrites the reduced gradient the previous one left) or applied
    twice (a stale bucket re-applied) while the race runs, and `pipelined` always leaves its last step's gradient in the
    bucket.  A training loop that cares races on throw-away steps, then calls `flush_schedule(name, bucket, lr)` once when
    it stops stepping (applies the trailing bucket of `pipelined`; a no-op for `three_part`)."""
    if forced is not None:
        if forced not in schemes:
            raise KeyError("unknown schedule %r (have: %s)" % (forced, ", ".join(schemes)))
        schemes = {forced: schemes[forced]}
    best, seen = None, {}
    for name, fns in schemes.items():
        run, captured = prepare(fns)
        for _ in range(warm):
            for f in run:
                f()
        if barrier is not None:
            barrier()
        sync()
        t0 = clock()
        for _ in range(timed):
            for f in run:
                f()
        sync()
        el = float(reduce_max(clock() - t0))
        seen[name] = el
        if best is None or el < best[0]:
            best = (el, name, run, captured)
    if best is None:
        raise ValueError("select_schedule: no schedule given")
    return best[1], best[2], best[3], seen


def flush_schedule(name, bucket, lr):
    """Finalizer of a run of `ddp_schedules` steps: `pipelined` applies step t's reduced gradient at the start of step
    t + 1, so the last step's is still in the bucket -- apply it and zero the bucket; `three_part` has nothing pending."""
    if name == "pipelined":
        bucket.sgd_step(lr)
        bucket.flat.zero_()


def prewarm_replays(run, prewarm_ms, sync, agree=None, clock=time.perf_counter, chunk=8):
    """Replay `run` (one whole step) for about `prewarm_ms` milliseconds of wall time before a timed region (bench.py: the
    GPU's clocks settle while the captured step is replayed).  With more than one rank a step holds a collective, so every
    rank has to leave this loop after the SAME number of steps: `agree(done) -> bool` turns one rank's "my time is up" into a
    decision all ranks share (bench.py broadcasts rank 0's) -- a loop that each rank ends on its own clock leaves the ranks
    with different numbers of all-reduces issued, i.e. a hang or gradients summed across different steps.
    Returns the number of steps replayed."""
    if prewarm_ms <= 0:
        return 0
    n = 0
    sync()
    t0 = clock()
    while True:
        for _ in range(chunk):
            run()
        n += chunk
        sync()
        done = (clock() - t0) * 1e3 >= prewarm_ms
        if agree is not None:
            done = bool(agree(done))
        if done:
            return n
