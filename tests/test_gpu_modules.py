"""-m gpu: product modules (HIP cores through the C ABI, bf16 autocast) vs the golden vectors
of the fp32 reference, forward and backward, eval and training mode (injected noise)."""
import pytest

import cases
from gpu_checks import check_module_case

READY = ("eva_", "local_", "lara_", "softmax_", "performer_", "causal_eva_", "ra_", "scatterbrain_")       # variants whose HIP cores have landed


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("mode", cases.MODES)
@pytest.mark.parametrize("name", [n for n in cases.CASES if n.startswith(READY)])
def test_module_matches_reference(name, mode, dtype):
    import torch
    errs = check_module_case(name, mode, dtype=torch.bfloat16 if dtype == "bf16" else torch.float16)
    print(name, mode, dtype, {k: "%.2e/%.2e" % v for k, v in errs.items()})


@pytest.mark.gpu
@pytest.mark.parametrize("mode", cases.MODES)
@pytest.mark.parametrize("name", ["lara_2d_dense_antithetic", "lara_2d_vmixed_bh", "lara_2d_poolmixed_tiny",
                                  "lara_2d_noparam_multisample"])
def test_lara_landmark_algebra_stays_on_hip(name, mode, monkeypatch):
    """The [L x L] / [C x L] landmark algebra of these variants (softmax mixing of k_bar, '-vmixed' column bias,
    proposal densities) runs in the HIP landmark kernels: no torch einsum / softmax / logsumexp may be reached."""
    import torch
    import torch.nn.functional as F

    def banned(*a, **k):
        raise AssertionError("torch einsum/softmax/logsumexp reached in the LARA landmark path")
    for mod, fn in ((torch, "einsum"), (torch, "softmax"), (torch, "logsumexp"), (F, "softmax"), (torch.Tensor, "softmax")):
        monkeypatch.setattr(mod, fn, banned)
    if "dense" in name:
        # round 4: the model-wide generator's LayerNorm runs on ea_layernorm_fwd / _bwd as well
        def no_ln(*a, **k):
            raise AssertionError("F.layer_norm reached in the 'dense' landmark generator")
        monkeypatch.setattr(F, "layer_norm", no_ln)
    errs = check_module_case(name, mode)
    print(name, mode, {k: "%.2e/%.2e" % v for k, v in errs.items()})
