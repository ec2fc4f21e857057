"""-m "not gpu": the whole-model harness (ea_harness) on CPU.

  * checkpoint compatibility: state_dict key / shape tables of DeiTStack and PvTStack == the
    reference's EfficientTransformer / PyramidVisionTransformerV2 (tests/golden/model_keys.json,
    dumped from the reference by tests/golden/gen_model_keys.py);
  * the N > 1 path on a stack: two gloo ranks, DistributedDataParallel through trainer.wrap_ddp /
    trainer.make_step, averaged half-batch gradients == full-batch gradients and identical parameters
    after the optimizer step.  The attention forward is stood in for by the oracle (the HIP cores
    need a GPU) -- every other line of the step is the harness's own."""
import json
import os
import socket
import sys
import warnings

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-attention_amd")]

import oracle  # noqa: E402

KEYS = json.load(open(os.path.join(ROOT, "tests", "golden", "model_keys.json")))


def _table(m):
    return {k: list(v.shape) for k, v in m.state_dict().items()}


def test_checkpoint_keys_match_reference_models():
    from ea_harness import trainer
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for wl_name, ref_name in (("model_cfg2", "deit_tiny_p16_eva"), ("model_cfg3", "deit_tiny_p8_lara"),
                                  ("model_cfg4", "pvt_b2_eva")):
            wl = trainer.build_workload(wl_name, torch.device("cpu"), batch=1)
            assert _table(wl.model) == KEYS[ref_name], wl_name


def test_encoder_adapter_parameter_names():
    from ea_harness import wmt_en_de_encoder
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = wmt_en_de_encoder("lara", dict(num_landmarks=16, proposal_gen="adaptive-1d"), vocab=64)
    ks = set(m.state_dict())
    for k in ("layers.0.self_attn.attn.qkv.weight", "layers.0.self_attn.attn.proj.bias",
              "layers.0.self_attn.attn.q_bar_gen.0.weight", "layers.5.final_layer_norm.weight", "layers.0.fc1.weight"):
        assert k in ks, k
    assert m.layers[0].self_attn.num_heads == 8 and m.layers[0].self_attn.head_dim == 64


ATTN = dict(window_size=7, attn_2d=True, use_rpe=True, num_landmarks=4, adaptive_proj="default")


def _oracle_backed(model, attn_name, attn_args):
    """Replace every block's attention forward by the oracle evaluated on the module's own parameters."""
    for blk in model.blocks:
        inner = blk.attn
        args = dict(attn_args, dim=inner.dim, num_heads=inner.num_heads)

        def fwd(x, key_padding_mask=None, inner=inner, args=args):
            params = dict(inner.named_parameters())
            params.update(dict(inner.named_buffers()))
            return oracle.module_forward(attn_name, args, params, x, key_padding_mask, training=False)
        inner.forward = fwd
    return model


def _build_stack():
    from ea_harness.vision import DeiTStack
    torch.manual_seed(3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = DeiTStack("eva", ATTN, img_size=56, patch_size=4, num_classes=10, embed_dim=64, depth=2, num_heads=2,
                      drop_path_rate=0.0)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.02 * torch.randn_like(p))
    return _oracle_backed(m, "eva", ATTN)


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ea_harness import trainer
        import torch.nn.functional as F
        torch.set_num_threads(2)
        model = _build_stack()
        torch.manual_seed(11)
        x = torch.randn(4, 3, 56, 56)
        y = torch.randint(0, 10, (4,))
        sl = slice(rank * 2, rank * 2 + 2)
        wl = trainer.Workload("tiny", model, x[sl], y[sl], lambda m, a, b: F.cross_entropy(m(a).float(), b), 2 * 196, "tiny")
        ddp = trainer.wrap_ddp(model, torch.device("cpu"))
        opt = torch.optim.SGD(model.parameters(), lr=0.1)
        step = trainer.make_step(wl, optimizer=opt, ddp_model=ddp)
        before = {k: p.detach().clone() for k, p in model.named_parameters()}
        step()
        for k, p in model.named_parameters():                       # the optimizer step used the averaged gradient
            assert torch.allclose(p.detach(), before[k] - 0.1 * p.grad, rtol=1e-6, atol=1e-7), k
        if rank == 0:
            ref = _build_stack()
            F.cross_entropy(ref(x).float(), y).backward()          # mean over the full batch == mean of the two half-batch means
            worst = 0.0
            for k, p in ref.named_parameters():
                assert p.grad is not None, k
                got = dict(model.named_parameters())[k].grad                                # averaged over the two ranks
                err = (got - p.grad).abs().max().item() / max(p.grad.abs().max().item(), 1e-12)
                worst = max(worst, err)
            ret["worst"] = worst
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_step_on_a_stack():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
        assert ret["worst"] < 2e-4, dict(ret)
