"""Data parallelism for the attention layers: one process per GPU, batch sharded, no data-path
collective (every (batch element, head) is independent -- SURVEY.md 8e).  The only exchange is the
parameter-gradient all-reduce, done here the MI355X way: ONE flat fp32 bucket per step (the layer
has ~0.16 M parameters, far below the size where xGMI ring bandwidth matters, so a single
latency-bound RCCL call beats per-bucket hooks), with the 1/world averaging folded into the
optimiser step.  Numerically this is DistributedDataParallel + SGD; unlike the DDP wrapper it adds
no per-iteration host work, so the forward/backward and the update can stay inside captured
hipGraphs with the all-reduce as the only eager call between them (bench.py)."""
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
