"""-m "not gpu": the drop-in surface -- factory names, constructor kwargs,
state_dict keys/shapes, buffers, argparse flags/defaults/prefix nesting -- and loud failure of the
attention cores without a GPU tensor (no CPU fallback)."""
import argparse
import warnings

import numpy as np
import pytest
import torch

import cases
import efficient_attention as ea
from util import Fixture


def build(case):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return ea.AttentionFactory.build_attention(case["attn"], cases.ctor_args(case))


@pytest.mark.parametrize("name", list(cases.CASES))
def test_state_dict_matches_reference(name):
    fx = Fixture(name)
    mod = build(fx.case)
    sd = mod.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == fx.key_shapes
    if "relative_position_index" in fx.z.files:
        assert np.array_equal(sd["relative_position_index"].numpy(), fx.z["relative_position_index"])
    # a reference checkpoint loads strictly
    new = {k: (torch.from_numpy(fx.params_np[k]) if k in fx.params_np else v) for k, v in sd.items()}
    mod.load_state_dict(new, strict=True)


def test_factory_names_and_errors():
    assert set(ea.AttentionFactory.attn_dict) == {"performer", "softmax", "local", "lara", "ra",
                                                  "scatterbrain", "eva", "causal_eva"}
    with pytest.raises(KeyError):
        ea.AttentionFactory.build_attention("nope", {})
    with pytest.raises(TypeError):
        ea.AttentionFactory.build_attention("softmax", dict(dim=64, num_heads=2, bogus=1))
    with pytest.raises(NotImplementedError):
        ea.AttentionFactory.build_attention("eva", dict(dim=64, num_heads=2, use_rpe=True, use_t5_rpe=True,
                                                        window_size=4))
    sb = ea.AttentionFactory.build_attention("scatterbrain", dict(dim=64, num_heads=2, window_size=4, overlap_window=True))
    assert sb.ext_size == 2                                      # window overlap: extended key patch (local_attention.py:36-40)


DEFAULTS = {
    "softmax": dict(fp32=False),
    "local": dict(fp32=False, use_rpe=False, window_size=4, attn_2d=False, overlap_window=False),
    "eva": dict(fp32=False, use_rpe=False, window_size=4, attn_2d=False, overlap_window=False,
                adaptive_proj="default", num_landmarks=49, use_t5_rpe=False),
    "lara": dict(fp32=False, num_landmarks=49, kernel_size=None, pool_module_type="light", mis_type="mis-opt",
                 proposal_gen="pool", use_antithetics=False, use_multisample=False, alpha_coeff=1.0),
    "performer": dict(fp32=False, approx_attn_dim=64, proj_method="favorp", cos_weighting=False,
                      sample_scheme="default"),
    "ra": dict(fp32=False, num_samples=1),
    "scatterbrain": dict(fp32=False, use_rpe=False, window_size=4, attn_2d=False, overlap_window=False,
                         approx_attn_dim=64, proj_method="favorp", cos_weighting=False, sample_scheme="default"),
}


@pytest.mark.parametrize("attn", list(DEFAULTS))
def test_argparse_defaults_and_nesting(attn):
    parser = argparse.ArgumentParser()
    parser = ea.AttentionFactory.add_attn_specific_args(parser, attn)
    args = parser.parse_args([], namespace=ea.NestedNamespace())
    assert vars(args.attn_args) == DEFAULTS[attn]
    # the kwargs the call sites build (efficient_vit.py:175-184) construct the module
    kw = dict(vars(args.attn_args), dim=64, num_heads=2, qkv_bias=True, attn_drop=0.0, proj_drop=0.0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ea.AttentionFactory.build_attention(attn, kw)


def test_argparse_prefix_like_fairseq():
    parser = argparse.ArgumentParser()
    ea.AttentionFactory.add_attn_specific_args(parser, "eva", struct_name="attn_args_encoder", prefix="encoder-attn")
    ea.AttentionFactory.add_attn_specific_args(parser, "lara", struct_name="attn_args_decoder", prefix="decoder-attn")
    args = parser.parse_args(["--encoder-attn-window-size", "8", "--encoder-attn-use-t5-rpe",
                              "--decoder-attn-alpha-coeff", "2.0"], namespace=ea.NestedNamespace())
    assert args.attn_args_encoder.window_size == 8 and args.attn_args_encoder.use_t5_rpe is True
    assert args.attn_args_encoder.num_landmarks == 49
    assert args.attn_args_decoder.alpha_coeff == 2.0 and args.attn_args_decoder.mis_type == "mis-opt"


def test_remove_argument_and_helpers():
    parser = argparse.ArgumentParser()
    ea.add_nested_argument(parser, "--window-size", default=3, type=int)
    ea.remove_argument(parser, "--window-size")
    assert parser.parse_args([], namespace=ea.NestedNamespace()).__dict__ == {}
    assert ea.remove_prefix("--enc-x", "--enc-") == "x" and ea.remove_prefix("abc", "zz") == "abc"


@pytest.mark.parametrize("attn", ["softmax", "local", "eva", "lara", "performer", "ra", "scatterbrain"])
def test_no_cpu_fallback(attn):
    """A CPU tensor must never be silently computed by something else."""
    args = dict(dim=64, num_heads=2)
    if attn in ("local", "eva", "scatterbrain"):
        args.update(window_size=4, num_landmarks=4) if attn == "eva" else args.update(window_size=4)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mod = ea.AttentionFactory.build_attention(attn, args).eval()
    x = torch.randn(2, 16, 64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mod(x)


CAUSAL_EVA_DEFAULTS = dict(adaptive_proj="default", num_chunks=None, chunk_size=None, causal=False,
                           use_t5_rpe=False, window_size=4, overlap_window=False)


def _causal_eva(**over):
    aa = dict(CAUSAL_EVA_DEFAULTS, adaptive_proj="qk", chunk_size=4, causal=True, window_size=8)
    aa.update(over.pop("attn_args", {}))
    kw = dict(embed_dim=64, num_heads=2, self_attention=True, attn_args=argparse.Namespace(**aa))
    kw.update(over)
    return ea.AttentionFactory.build_attention("causal_eva", kw)


def test_causal_eva_flags_and_loud_failures():
    """causal_eva.py:905-916 flag defaults with fairseq's decoder prefix; the paths this build
    does not carry (quantization noise, decoding without the causal masks) raise instead of
    computing something else, and CPU tensors never reach a kernel."""
    parser = argparse.ArgumentParser()
    parser = ea.AttentionFactory.add_attn_specific_args(parser, "causal_eva", struct_name="attn_args_decoder",
                                                        prefix="decoder-attn")
    args = parser.parse_args([], namespace=ea.NestedNamespace())
    assert vars(args.attn_args_decoder) == CAUSAL_EVA_DEFAULTS
    args = parser.parse_args("--decoder-attn-window-size 128 --decoder-attn-causal --decoder-attn-adaptive-proj qk "
                             "--decoder-attn-chunk-size 8 --decoder-attn-use-t5-rpe".split(),
                             namespace=ea.NestedNamespace())
    mod = ea.CausalEVAttention(1024, 8, dropout=0.1, self_attention=True, attn_args=args.attn_args_decoder)
    assert (mod.num_heads, mod.head_dim, mod.window_size, mod.ext_size, mod.chunk_size) == (8, 128, 128, 0, 8)
    assert mod.rel_pos_bias.relative_attention_bias.weight.shape == (64, 1) and mod.rel_pos_bias.causal
    x = torch.randn(16, 2, 64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _causal_eva().eval()(x, x, x)
    # incremental decoding is a HIP path too (tests/test_gpu_causal_eva.py): a CPU tensor fails loudly, and the cases the
    # build cannot pin against the full-sequence path (no --causal, training mode) raise
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _causal_eva().eval()(x[:1], x[:1], x[:1], incremental_state={})
    with pytest.raises(NotImplementedError, match="incremental"):
        _causal_eva(attn_args=dict(causal=False)).eval()(x[:1], x[:1], x[:1], incremental_state={})
    with pytest.raises(NotImplementedError, match="incremental"):
        _causal_eva().train()(x[:1], x[:1], x[:1], incremental_state={})
    # quantization noise: the reference's size check (causal_eva.py:149-151) and its weight update (:165-213) -- the noise
    # itself is parameter arithmetic, testable without a GPU
    with pytest.raises(AssertionError, match="multiple of block sizes"):
        _causal_eva(q_noise=0.1, qn_block_size=7)
    qn = _causal_eva(q_noise=0.25, qn_block_size=8).train()
    w0 = qn.q_proj.weight.detach().clone()
    drop = (torch.arange(w0.numel() // 8) % 3 == 0).float()
    qn._qnoise_mask_fn = lambda n: drop[:n]
    qn._quant_noise_(qn.q_proj)
    blocks = drop.bool().repeat_interleave(8).view_as(w0)
    assert torch.equal(qn.q_proj.weight.detach(), torch.where(blocks, torch.zeros_like(w0), w0 * (1 / 0.75)))
    qn.eval()._quant_noise_(qn.k_proj)                          # evaluation mode: no noise
    qn._qnoise_mask_fn = None
    torch.manual_seed(0)
    k0 = qn.k_proj.weight.detach().clone()
    qn.train()._quant_noise_(qn.k_proj)
    zero_blocks = (qn.k_proj.weight.detach().view(-1, 8) == 0).all(-1)
    assert 0.15 < zero_blocks.float().mean() < 0.35
    kept = ~zero_blocks.repeat_interleave(8).view_as(k0)
    assert torch.allclose(qn.k_proj.weight.detach()[kept], k0[kept] / 0.75)
    with pytest.raises(AssertionError):
        _causal_eva(attn_args=dict(chunk_size=3))               # window % chunk != 0 (causal_eva.py:362)
    # legacy fused in_proj checkpoints are split like the reference does (:876-903)
    sd = {"m.in_proj_weight": torch.arange(12.).view(6, 2), "m.in_proj_bias": torch.arange(6.)}
    _causal_eva().upgrade_state_dict_named(sd, "m")
    assert sorted(sd) == ["m.k_proj.bias", "m.k_proj.weight", "m.q_proj.bias", "m.q_proj.weight",
                          "m.v_proj.bias", "m.v_proj.weight"]
    assert torch.equal(sd["m.k_proj.weight"], torch.arange(12.).view(6, 2)[2:4])


def test_product_package_does_not_import_oracle():
    import os
    pkg = os.path.dirname(ea.__file__)
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn
