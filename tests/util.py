"""Shared helpers for the parity tests: fixture loading and tolerant comparisons."""
import json
import os

import numpy as np
import torch

import cases  # tests/golden/cases.py

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Fixture:
    def __init__(self, name):
        self.name = name
        self.case = cases.CASES[name]
        self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.key_shapes = json.loads(str(self.z["key_shapes"]))
        self.params_np = cases.make_params(name, self.key_shapes)
        self.x_np, self.g_np, self.mask_np = cases.make_inputs(name)

    def params(self, dtype=torch.float32, device="cpu", requires_grad=False):
        out = {}
        for k, v in self.params_np.items():
            t = torch.from_numpy(v).to(device=device, dtype=dtype)
            out[k] = t.requires_grad_(requires_grad)
        return out

    def noise_fn(self, mode, dtype=torch.float32, device="cpu"):
        """i-th call -> the same stream the generator fed to the reference."""
        calls = []

        def fn(shape):
            arr = cases.make_noise(self.name, tuple(shape), len(calls))
            calls.append(tuple(shape))
            return torch.from_numpy(arr).to(device=device, dtype=dtype)

        fn.calls = calls
        return fn

    def keep_fn(self, dtype=torch.float32, device="cpu"):
        """i-th attention-dropout call -> the keep decisions the generator fed to the reference."""
        calls = []
        p_drop = float(self.case["args"].get("dropout", self.case["args"].get("attn_drop", 0.0)))

        def fn(shape):
            arr = cases.make_keep(self.name, tuple(shape), p_drop, len(calls))
            calls.append(tuple(shape))
            return torch.from_numpy(arr).to(device=device, dtype=dtype)

        fn.calls = calls
        return fn

    def qmask_fn(self, device="cpu"):
        """i-th quantization-noise draw -> the block-drop decisions the generator fed to the reference."""
        calls = []
        p = float(self.case["args"].get("q_noise", 0.0))

        def fn(n):
            arr = cases.make_block_mask(self.name, int(n), p, len(calls))
            calls.append(int(n))
            return torch.from_numpy(arr).to(device)

        fn.calls = calls
        return fn

    def expected_qn_blocks(self, mode):
        key = "%s.qn_blocks" % mode
        return json.loads(str(self.z[key])) if key in self.z.files else []

    def index_fn(self, device="cpu"):
        """i-th multinomial call of randomized attention -> the draws the generator fed to the reference
        (there as [B*h*N, 1] of N-way draws, here as [B,h,N])."""
        calls = []

        def fn(shape):
            n = int(np.prod(shape))
            arr = cases.make_index(self.name, (n, 1), shape[-1], len(calls)).reshape(shape)
            calls.append(tuple(shape))
            return torch.from_numpy(arr).to(device)

        fn.calls = calls
        return fn

    def expected_drop_elems(self, mode):
        key = "%s.drop_shapes" % mode
        if key not in self.z.files:
            return []
        return [int(np.prod(s)) for s in json.loads(str(self.z[key]))]

    def expected_noise_shapes(self, mode):
        return [tuple(s) for s in json.loads(str(self.z["%s.noise_shapes" % mode]))]

    def y(self, mode):
        return self.z["%s.y" % mode]

    def dx(self, mode):
        return self.z["%s.dx" % mode]

    def grad_keys(self, mode):
        pre = "%s.grad." % mode
        keys = set()
        for k in self.z.files:
            if k.startswith(pre):
                k = k[len(pre):]
                for suf in (".sample", ".moments"):
                    if k.endswith(suf):
                        k = k[:-len(suf)]
                keys.add(k)
        return sorted(keys)

    def check_grad(self, mode, key, got, rtol, atol):
        """Compare a parameter gradient with the stored full array or subsample+moments."""
        got = np.asarray(got, np.float64)
        pre = "%s.grad.%s" % (mode, key)
        if pre in self.z.files:
            ref = self.z[pre].astype(np.float64)
            assert got.shape == ref.shape, (key, got.shape, ref.shape)
            assert_close(got, ref, rtol, atol, "%s/%s grad %s" % (self.name, mode, key))
            return
        idx = cases.grad_sample_index(self.name, key, got.size)
        ref = self.z[pre + ".sample"].astype(np.float64)
        assert_close(got.reshape(-1)[idx], ref, rtol, atol, "%s/%s grad %s[sample]" % (self.name, mode, key))
        mom = self.z[pre + ".moments"]
        flat = got.reshape(-1)
        # moments are sums over up to ~1e6 entries: compare relative to the abs-sum
        assert abs(flat.sum() - mom[0]) <= rtol * 50 * mom[1] + atol, (key, flat.sum(), mom)
        assert abs(np.abs(flat).sum() - mom[1]) <= rtol * 50 * mom[1] + atol, (key,)


def assert_close(got, ref, rtol, atol, what=""):
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    err = np.abs(got - ref)
    tol = atol + rtol * np.abs(ref)
    bad = err > tol
    if bad.any():
        i = np.argmax(err - tol)
        raise AssertionError("%s: %d/%d out of tolerance; worst |err|=%.3e at flat %d (got %.6e ref %.6e), "
                             "max|ref|=%.3e" % (what, bad.sum(), bad.size, err.reshape(-1)[i], i,
                                                got.reshape(-1)[i], ref.reshape(-1)[i], np.abs(ref).max()))


def elementwise_excess(got, ref, coef):
    """max over the elements of |got - ref| / (coef[0] * rms(ref) + coef[1] * |ref|): <= 1 <=> EVERY element obeys
    |err| <= coef[0] rms(ref) + coef[1] |ref|.  The norm-wise figures of scaled_err() let one wrong small-magnitude channel
    hide under max|ref|; this one does not."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    rms = float(np.sqrt((ref * ref).mean()))
    ex = float((np.abs(got - ref) / np.maximum(coef[0] * rms + coef[1] * np.abs(ref), 1e-300)).max())
    log = os.environ.get("EA_TEST_ERR_LOG")
    if log:
        with open(log + ".elem", "a") as f:
            f.write("%s\t%.6g\t%.6g\t%.6g\n" % (os.environ.get("PYTEST_CURRENT_TEST", "?"), coef[0], coef[1], ex))
    return ex


def scaled_err(got, ref):
    """max |got-ref| / max|ref| and rms(got-ref)/rms(ref): the two figures the bf16 tolerances
    are stated in."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    d = got - ref
    out = np.abs(d).max() / max(np.abs(ref).max(), 1e-30), np.sqrt((d * d).mean()) / max(np.sqrt((ref * ref).mean()), 1e-30)
    log = os.environ.get("EA_TEST_ERR_LOG")          # dev: collect the observed errors (tools/tol_report.py sets the bounds from them)
    if log:
        with open(log, "a") as f:
            f.write("%s\t%.6g\t%.6g\n" % (os.environ.get("PYTEST_CURRENT_TEST", "?"), out[0], out[1]))
    return out
