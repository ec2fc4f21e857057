"""Pin the oracle: CPU restatement vs the reference's own outputs (tests/golden/*.npz,
produced by tests/golden/gen_golden.py from /root/reference).  fp32, rtol 1e-4 / atol 2e-5
forward and 2e-4 / 1e-4 on gradients (the reference and the restatement order their fp32
reductions differently)."""
import numpy as np
import pytest
import torch

import cases
import oracle
from util import Fixture, assert_close


@pytest.mark.parametrize("mode", cases.MODES)
@pytest.mark.parametrize("name", list(cases.CASES))
def test_oracle_matches_reference(name, mode):
    fx = Fixture(name)
    case = fx.case
    params = fx.params(requires_grad=True)
    x = torch.from_numpy(fx.x_np).clone().requires_grad_(True)
    mask = None if fx.mask_np is None else torch.from_numpy(fx.mask_np)
    noise_fn, keep_fn, qmask_fn = fx.noise_fn(mode), fx.keep_fn(), fx.qmask_fn()
    y = oracle.module_forward(case["attn"], case["args"], params, x, mask,
                              training=(mode == "train"), noise_fn=noise_fn, keep_fn=keep_fn,
                              index_fn=fx.index_fn(), qmask_fn=qmask_fn)
    assert noise_fn.calls == fx.expected_noise_shapes(mode)
    assert qmask_fn.calls == fx.expected_qn_blocks(mode)
    assert [int(np.prod(s)) for s in keep_fn.calls] == fx.expected_drop_elems(mode)
    assert_close(y.detach().numpy(), fx.y(mode), 1e-4, 2e-5, "%s/%s y" % (name, mode))
    (y * torch.from_numpy(fx.g_np)).sum().backward()
    assert_close(x.grad.numpy(), fx.dx(mode), 2e-4, 1e-4, "%s/%s dx" % (name, mode))
    for key in fx.grad_keys(mode):
        g = params[key].grad
        g = np.zeros(tuple(params[key].shape), np.float32) if g is None else g.numpy()
        fx.check_grad(mode, key, g, 2e-4, 1e-4)


@pytest.mark.parametrize("name", [n for n in cases.CASES if "relative_position_index" in Fixture(n).z.files])
def test_rpe_index_matches_reference(name):
    fx = Fixture(name)
    a = fx.case["args"]
    w = a["window_size"]
    e = max(1, w // 2) if a.get("overlap_window", False) else 0
    assert np.array_equal(oracle.rpe_index_2d(w, e).numpy(), fx.z["relative_position_index"])
