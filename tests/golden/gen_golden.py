#!/usr/bin/env python3
"""Generate golden vectors for the attention hot path by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference).  Nothing of the reference is
copied: this script imports it, feeds it the numpy-seeded inputs/parameters/noise defined in
cases.py, and stores the reference's OUTPUTS (y, dL/dx, dL/dtheta for L = sum(y*g)) as
.npz fixtures next to this file.  The tests then compare oracle/ (CPU) and the HIP path
(GPU) against these files without ever touching /root/reference.

    python tests/golden/gen_golden.py            # regenerate everything
    python tests/golden/gen_golden.py eva_2d_rpe_tiny lara_1d_even

`timm` is not installed here; the reference only uses timm.models.layers.trunc_normal_
(abstract_attention.py:5, local_attention.py:10), which is shimmed to
torch.nn.init.trunc_normal_ in a throw-away module placed on sys.path at run time.
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases  # noqa: E402

REF_ROOT = "/root/reference/efficient-attention"


def _import_reference():
    shim_dir = tempfile.mkdtemp(prefix="timm_shim_")
    timm = types.ModuleType("timm")
    models = types.ModuleType("timm.models")
    layers = types.ModuleType("timm.models.layers")
    layers.trunc_normal_ = torch.nn.init.trunc_normal_
    timm.models = models
    models.layers = layers
    sys.modules["timm"] = timm
    sys.modules["timm.models"] = models
    sys.modules["timm.models.layers"] = layers
    sys.path.insert(0, REF_ROOT)
    import efficient_attention as ref  # the reference package
    assert ref.__file__.startswith(REF_ROOT), ref.__file__
    return ref


class _NoisePatch:
    """Replace torch.randn / torch.randn_like during one forward with cases.make_noise, and
    F.dropout (causal_eva's attention dropout, causal_eva.py:227) with cases.make_keep decisions."""

    def __init__(self, name):
        self.name = name
        self.calls = []
        self.drops = []

    def __enter__(self):
        self._randn, self._randn_like = torch.randn, torch.randn_like
        self._dropout = torch.nn.functional.dropout

        def dropout(x, p=0.5, training=True, inplace=False):
            if not training or p == 0:
                return x
            keep = torch.from_numpy(cases.make_keep(self.name, tuple(x.shape), p, len(self.drops)))
            self.drops.append(tuple(x.shape))
            return x * keep.reshape(x.shape).to(x.dtype) / (1.0 - p)

        torch.nn.functional.dropout = dropout
        self._multinomial = torch.multinomial
        self.draws = []

        def multinomial(probs, num_samples, replacement=False, **kw):
            # RA: torch.multinomial(pi.reshape(b*h*n, n), 1, replacement=True) (randomized_attention.py:36)
            assert num_samples == 1
            idx = cases.make_index(self.name, (probs.shape[0], 1), probs.shape[1], len(self.draws))
            self.draws.append(tuple(probs.shape))
            return torch.from_numpy(idx)

        torch.multinomial = multinomial

        def randn(*size, **kw):
            if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
                size = tuple(size[0])
            arr = cases.make_noise(self.name, size, len(self.calls))
            self.calls.append(tuple(size))
            return torch.from_numpy(arr).to(kw.get("dtype") or torch.float32)

        def randn_like(t, **kw):
            arr = cases.make_noise(self.name, tuple(t.shape), len(self.calls))
            self.calls.append(tuple(t.shape))
            return torch.from_numpy(arr).to(t.dtype)

        torch.randn, torch.randn_like = randn, randn_like
        # quantization noise (causal_eva.py:175-179): `mask = torch.zeros(n); mask.bernoulli_(p)` per projection
        self._bernoulli_ = torch.Tensor.bernoulli_
        self.qn = []

        def bernoulli_(t, p=0.5, **kw):
            assert t.dim() == 1
            arr = cases.make_block_mask(self.name, t.numel(), float(p), len(self.qn))
            self.qn.append(int(t.numel()))
            return t.copy_(torch.from_numpy(arr))

        torch.Tensor.bernoulli_ = bernoulli_
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randn_like = self._randn, self._randn_like
        torch.nn.functional.dropout = self._dropout
        torch.multinomial = self._multinomial
        torch.Tensor.bernoulli_ = self._bernoulli_


def run_case(ref, name):
    case = cases.CASES[name]
    torch.manual_seed(0)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mod = ref.AttentionFactory.build_attention(case["attn"], cases.ctor_args(case))
    sd = mod.state_dict()
    key_shapes = {k: list(v.shape) for k, v in sd.items()}
    params = cases.make_params(name, key_shapes)
    new_sd = {}
    for k, v in sd.items():
        new_sd[k] = v if k not in params else torch.from_numpy(params[k]).to(v.dtype)
    mod.load_state_dict(new_sd, strict=True)

    x_np, g_np, mask_np = cases.make_inputs(name)
    out = {"key_shapes": np.array(json.dumps(key_shapes))}
    if "relative_position_index" in sd:
        out["relative_position_index"] = sd["relative_position_index"].numpy()

    for mode in cases.MODES:
        mod.train(mode == "train")
        mod.zero_grad(set_to_none=True)
        x = torch.from_numpy(x_np).clone().requires_grad_(True)
        mask = None if mask_np is None else torch.from_numpy(mask_np)
        with _NoisePatch(name) as np_patch:
            y = cases.call_module(case, mod, x, mask)
        (y * torch.from_numpy(g_np)).sum().backward()
        assert torch.isfinite(y).all(), (name, mode)
        out["%s.y" % mode] = y.detach().numpy()
        out["%s.dx" % mode] = x.grad.numpy()
        out["%s.noise_shapes" % mode] = np.array(json.dumps(np_patch.calls))
        out["%s.drop_shapes" % mode] = np.array(json.dumps(np_patch.drops))
        out["%s.draw_shapes" % mode] = np.array(json.dumps(np_patch.draws))
        if np_patch.qn:
            out["%s.qn_blocks" % mode] = np.array(json.dumps(np_patch.qn))
        for k, p in mod.named_parameters():
            gnp = np.zeros(tuple(p.shape), np.float32) if p.grad is None else p.grad.numpy()
            for suffix, arr in cases.pack_grad(name, k, gnp).items():
                out["%s.grad.%s%s" % (mode, k, suffix)] = arr
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    return path, os.path.getsize(path)


def main(argv):
    ref = _import_reference()
    names = argv or list(cases.CASES)
    for name in names:
        path, size = run_case(ref, name)
        print("%-34s %8.1f KB" % (name, size / 1024.0))


if __name__ == "__main__":
    main(sys.argv[1:])
