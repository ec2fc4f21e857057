"""GPU-side parity helpers shared by the -m gpu tests and __graft_entry__.smoke().

Module-level check: the product nn.Module (HIP cores, bf16 autocast like the reference's AMP
recipe) vs the golden vectors of the fp32 reference; core-level checks compare a HIP core with
the oracle evaluated in fp64 on the same bf16-rounded inputs.

Stated tolerances (bf16 operands: 8 significant bits, unit round-off 2^-9 = 2e-3; fp32
accumulation; every figure is relative to the reference tensor's own scale):
  module level vs fp32 golden : max|err| <= 4e-2 * max|ref|  and  rms(err) <= 2e-2 * rms(ref)
  LARA module level           : max|err| <= 5e-2 * max|ref|  and  rms(err) <= 2.5e-2 * rms(ref)
      (its gradients pass through three more bf16-rounded stages than a softmax kernel: the
       importance weights, d(log alpha)/alpha and the softmax-over-sequence term t(dt - u), whose
       two halves cancel, and the landmark matrices are rounded to bf16 exactly as autocast
       rounds einsum operands; observed <= 2.3e-2, 8x smaller in fp16)
  ScatterBrain module level  : max|err| <= 8e-2 * max|ref|  and  rms(err) <= 4e-2 * rms(ref)   (bf16 only)
      (its feature half multiplies bf16-rounded exponentials five deep -- phi(k), the joint weights A, the
       window statistics KV, dR, d phi(k) -- through three softmax-like normalisations; observed <= 6.6e-2 /
       3.3e-2 against the fp32 reference, and 1e-3 in fp16, where it shares the common bound below: the
       deviation is operand rounding; the kernels agree with an fp32 evaluation of the same bf16 qkv to
       6e-3, tools/sb_check.py)
  fp16 autocast (the reference's own AMP dtype, 11 significant bits), every variant:
                                max|err| <= 1e-2 * max|ref|  and  rms(err) <= 5e-3 * rms(ref)
  core level vs fp64 oracle   : max|err| <= 2e-2 * max|ref|  and  rms(err) <= 1e-2 * rms(ref)
"""
import contextlib
import warnings

import numpy as np
import torch

import cases
from util import Fixture, scaled_err, elementwise_excess

# Class ceilings (max |err| / max |ref|, rms err / rms ref) -- what DESIGN.md states for bf16 / fp16 operands.  The bound a
# test actually applies is tol_for(): ~2x the worst error OBSERVED for that (test file, variant, dtype) on MI355X
# (tests/golden/observed_errors.json, collected with EA_TEST_ERR_LOG by tools/tol_report.py), never
# above the ceiling -- so a regression of a factor two fails instead of hiding under a generous class bound.
MODULE_TOL = (4e-2, 2e-2)
LARA_TOL = (5e-2, 2.5e-2)
SCATTER_TOL = (8e-2, 4e-2)
FP16_TOL = (1e-2, 5e-3)
CORE_TOL = (2e-2, 1e-2)
# Cases with a stated bound of their own.  performer_2d_clamp: about half of the queries sit under the clamp of the
# normaliser (kernelized_attention.py:55), where the derivative is DISCONTINUOUS -- a query whose denominator lies within bf16
# rounding of 1e-2 takes the other branch than the fp32 reference and its whole gradient row differs.  y agrees to 6e-3;
# the gradients are bounded at 2x their observed error (0.069 / 0.033); the fp16 run of the same case keeps the common bound.
# Round 4: the Performer core itself now computes in exact fp32 arithmetic (ea_performer_f32_*) -- the flips that remain under
# bf16 autocast come from the bf16 rounding of q, k by the qkv projection (the same 0.069 / 0.033 with either core); in fp32
# outside autocast the case matches the reference at 2e-4 / 1e-4 (tests/test_gpu_performer_f32.py).
CASE_TOL = {("performer_2d_clamp", "bf16"): (1.4e-1, 7e-2)}
# ELEMENT-WISE bounds (round 5): every element obeys |err| <= a * rms(ref) + b * |ref| with (a, b) below --
# the norm-wise bounds above scale by max|ref| and would let a single wrong small-magnitude output channel pass.  fp16 is the
# round-2 bound of tests/test_gpu_fullsize.py; bf16 operands carry 8x its unit round-off.  Cases whose gradient is
# DISCONTINUOUS in the operands (the Performer clamp fixture) are exempt (None).
# Observed on MI355X (round 5, whole suite, EA_TEST_ERR_LOG -> *.elem): worst excess 0.61 of the bf16 bound (eva_2d_overlap_rpe),
# 0.34 of the fp16 bound; LARA's gradients (three more rounded stages, see the module docstring) reach 0.098 rms + 0.098 |ref|
# in bf16 and 0.015 / 0.015 in fp16 (lara_2d_dense_antithetic), ScatterBrain 0.068 / 0.014: their bounds are ~1.5x those.
ELEM_TOL = {"fp16": (1.5e-2, 1.5e-2), "bf16": (8e-2, 8e-2)}
# Performer (bf16 q, k into an exponential feature map): 0.079 / 0.079 observed (performer_1d_mask).  ScatterBrain's input
# gradient is heavy-tailed (max|ref| ~ 7.5 rms in scatterbrain_2d) and its rounding error is that of the row's LARGE entries
# spread over every channel by the projection product: elements with a small reference value carry errors of 0.44 rms in bf16
# (0.033 rms in fp16) while the norm-wise figures stay at 5.9e-2 / 3.0e-2 -- its bound says so instead of pretending otherwise.
ELEM_TOL_VARIANT = {("lara", "bf16"): (1.5e-1, 1.5e-1), ("lara", "fp16"): (2.5e-2, 2.5e-2),
                    ("performer", "bf16"): (1.2e-1, 1.2e-1),
                    # ScatterBrain in bf16 is checked NORM-WISE ONLY (SCATTER_TOL above): a (0.7, 0.7) "bound" admits a 70 % wrong
                    # element and was a bound in name only.  Its element-wise evidence is the fp16 run of the
                    # same fixtures (5e-2 / 5e-2) and tools/sb_check.py (kernels vs an fp32 evaluation of the same bf16 qkv: 6e-3)
                    ("scatterbrain", "bf16"): None, ("scatterbrain", "fp16"): (5e-2, 5e-2)}
ELEM_EXEMPT = {("performer_2d_clamp", "bf16")}


def elem_tol_for(attn, dtype, name=None):
    if (name, dtype) in ELEM_EXEMPT:
        return None
    if (attn, dtype) in ELEM_TOL_VARIANT:
        return ELEM_TOL_VARIANT[(attn, dtype)]            # (None: no element-wise bound for this variant / dtype)
    return ELEM_TOL[dtype]
_OBSERVED = None


def class_tol(attn, dtype):
    if dtype == "fp16":
        return FP16_TOL
    return {"lara": LARA_TOL, "scatterbrain": SCATTER_TOL}.get(attn, MODULE_TOL)


def tol_for(attn, dtype, where):
    """(max, rms) bound for variant `attn`, dtype 'bf16' | 'fp16', in test file `where` (its stem)."""
    global _OBSERVED
    import json
    import os
    if _OBSERVED is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "observed_errors.json")
        _OBSERVED = json.load(open(path)) if os.path.exists(path) else {}
    ceil = class_tol(attn, dtype)
    obs = _OBSERVED.get("%s|%s|%s" % (where, attn, dtype))
    if obs is None or os.environ.get("EA_TEST_ERR_LOG"):      # (while collecting: the class ceilings)
        return ceil
    floor = (4e-3, 3e-3) if dtype == "bf16" else (1e-3, 1e-3)
    return (min(ceil[0], max(2.0 * obs[0], floor[0])), min(ceil[1], max(2.0 * obs[1], floor[1])))


@contextlib.contextmanager
def injected_noise(fx, mode, device):
    """Route torch.randn / torch.randn_like to the fixture's noise streams during a forward."""
    calls = []
    real_randn, real_like = torch.randn, torch.randn_like

    def randn(*size, **kw):
        if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
            size = tuple(size[0])
        arr = cases.make_noise(fx.name, tuple(size), len(calls))
        calls.append(tuple(size))
        return torch.from_numpy(arr).to(device=kw.get("device", device), dtype=kw.get("dtype") or torch.float32)

    def randn_like(t, **kw):
        arr = cases.make_noise(fx.name, tuple(t.shape), len(calls))
        calls.append(tuple(t.shape))
        return torch.from_numpy(arr).to(device=t.device, dtype=t.dtype)

    torch.randn, torch.randn_like = randn, randn_like
    try:
        yield calls
    finally:
        torch.randn, torch.randn_like = real_randn, real_like


def build_module(fx, device="cuda"):
    import efficient_attention as ea
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mod = ea.AttentionFactory.build_attention(fx.case["attn"], cases.ctor_args(fx.case))
    sd = mod.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == fx.key_shapes
    new = {k: (torch.from_numpy(fx.params_np[k]) if k in fx.params_np else v) for k, v in sd.items()}
    mod.load_state_dict(new, strict=True)
    return mod.to(device)


def check_module_case(name, mode, backward=True, dtype=torch.bfloat16, tol=None):
    fx = Fixture(name)
    if tol is None:
        dts = "fp16" if dtype == torch.float16 else "bf16"
        tol = CASE_TOL.get((name, dts)) or tol_for(fx.case["attn"], dts, "test_gpu_modules")
    mod = build_module(fx)
    mod.train(mode == "train")
    keep_fn = fx.keep_fn(device="cuda")
    if hasattr(mod, "_keep_mask_fn"):
        mod._keep_mask_fn = keep_fn                      # attention dropout: the fixture's decisions
    qmask_fn = fx.qmask_fn(device="cuda")
    if hasattr(mod, "_qnoise_mask_fn"):
        mod._qnoise_mask_fn = qmask_fn                   # quantization noise: the fixture's block drops
    if hasattr(mod, "_sample_index_fn"):
        mod._sample_index_fn = fx.index_fn(device="cuda")  # randomized attention: the fixture's draws
    x = torch.from_numpy(fx.x_np).cuda().requires_grad_(True)
    mask = None if fx.mask_np is None else torch.from_numpy(fx.mask_np).cuda()
    with injected_noise(fx, mode, "cuda") as calls:
        # dtype = torch.float32: NO autocast -- the module called the way the reference computes in fp32
        with (torch.autocast("cuda", dtype=dtype) if dtype != torch.float32 else contextlib.nullcontext()):
            y = cases.call_module(fx.case, mod, x, mask)
    assert calls == fx.expected_noise_shapes(mode), (calls, fx.expected_noise_shapes(mode))
    assert [int(np.prod(s)) for s in keep_fn.calls] == fx.expected_drop_elems(mode)
    assert qmask_fn.calls == fx.expected_qn_blocks(mode)
    assert y.shape == x.shape and y.dtype in (dtype, torch.float32)
    elem = {}

    def both(key, got, ref):
        errs[key] = scaled_err(got, ref)
        elem[key] = lambda coef, got=got, ref=ref: elementwise_excess(got, ref, coef)
    errs = {}
    both("y", y.detach().float().cpu().numpy(), fx.y(mode))
    if backward:
        (y.float() * torch.from_numpy(fx.g_np).cuda()).sum().backward()
        both("dx", x.grad.float().cpu().numpy(), fx.dx(mode))
        for key, p in mod.named_parameters():
            pre = "%s.grad.%s" % (mode, key)
            if pre in fx.z.files:
                ref = fx.z[pre]
                got = np.zeros_like(ref) if p.grad is None else p.grad.float().cpu().numpy()
                if np.abs(ref).max() > 0:
                    both("d" + key, got, ref)
                else:
                    assert np.abs(got).max() == 0, key
            else:
                idx = cases.grad_sample_index(name, key, p.numel())
                ref = fx.z[pre + ".sample"]
                got = p.grad.float().cpu().numpy().reshape(-1)[idx]
                both("d" + key, got, ref)
    bad = {k: v for k, v in errs.items() if not (v[0] <= tol[0] and v[1] <= tol[1])}
    assert not bad, "%s/%s out of tolerance %s: %s (all: %s)" % (name, mode, tol, bad, errs)
    etol = None if dtype == torch.float32 else elem_tol_for(fx.case["attn"], "fp16" if dtype == torch.float16 else "bf16", name)
    if etol is not None:
        ebad = {k: v for k, v in elem.items() if v(etol) > 1.0}
        assert not ebad, "%s/%s element-wise bound %s exceeded: %s" % (name, mode, etol, {k: v(etol) for k, v in ebad.items()})
    return errs
