"""-m gpu: the whole-model harness (ea_harness) through the HIP attention cores.

  * every BASELINE.json model workload (configs 2-5) takes one optimizer step at a reduced batch: finite
    loss, every parameter receives a finite gradient (DistributedDataParallel's requirement);
  * the Time x Batch x Channel adapter returns exactly the batch-first module's output, transposed;
  * a captured hipGraph replay of the step reproduces the eager step's loss trajectory."""
import os
import sys
import warnings

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-attention_amd")]

BATCH = {"model_cfg2": 8, "model_cfg3": 4, "model_cfg4": 2, "model_cfg5": 1}


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(BATCH))
def test_model_workload_takes_a_step(name):
    from ea_harness import trainer
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        wl = trainer.build_workload(name, torch.device("cuda"), batch=BATCH[name])
    step = trainer.make_step(wl)
    l0 = float(step())
    for k, p in wl.model.named_parameters():
        assert p.grad is not None, k
        assert torch.isfinite(p.grad).all(), k
    l1 = float(step())
    assert l0 == l0 and l1 == l1 and abs(l0) < 1e4


@pytest.mark.gpu
def test_time_first_adapter_is_a_transpose():
    from ea_harness.sequence import TimeFirstSelfAttention
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = TimeFirstSelfAttention(512, 8, "lara", dict(num_landmarks=16, proposal_gen="adaptive-1d")).cuda().eval()
    x = torch.randn(300, 3, 512, device="cuda")
    mask = torch.zeros(3, 300, dtype=torch.bool, device="cuda")
    mask[1, 250:] = True
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y, w = m(x, x, x, key_padding_mask=mask)
        ref = m.attn(x.transpose(0, 1), mask)
    assert w is None and y.shape == x.shape
    assert torch.equal(y, ref.transpose(0, 1))


@pytest.mark.gpu
def test_captured_step_matches_eager():
    from ea_harness import trainer

    def run(graph):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            wl = trainer.build_workload("model_cfg2", torch.device("cuda"), batch=4, seed=5)
        wl.model.eval()                       # no sampling noise / stochastic depth: the two runs are comparable
        for p in wl.model.parameters():
            p.requires_grad_(True)
        step = trainer.make_step(wl, optimizer=torch.optim.SGD(wl.model.parameters(), lr=0.05))
        if graph:
            fn = trainer.capture_step(step, warmup=1)       # one eager step, then the capture (records, does not run)
        else:
            fn = step
            step()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        return [p.detach().float().clone() for p in wl.model.parameters()]
    a, b = run(False), run(True)
    for pa, pb in zip(a, b):
        assert torch.allclose(pa, pb, rtol=2e-2, atol=2e-3)
