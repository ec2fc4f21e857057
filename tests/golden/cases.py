"""Golden-vector case table shared by the generator (gen_golden.py, runs only in the
build container where /root/reference exists) and by the tests (which never touch the
reference).

Every tensor that goes INTO a case -- parameters, inputs, sampling noise, the cotangent
`g` -- is drawn from numpy's PCG64 with a seed derived from the case name, so the
fixtures only have to store what comes OUT of the reference (y, dx, parameter grads) plus
the parameter key/shape table.  numpy's bit generators are stable across platforms, so
the GPU box regenerates exactly the same inputs.
"""
import argparse
import zlib
import numpy as np

# name -> dict(attn=..., args=..., x_shape=..., mask=..., modes=(...))
# `mask`: None, or ("tail", [n_pad per batch row]) -> key_padding_mask with trailing pads.
CASES = {
    # ---------------- EVA (efficient_attention/eva.py:69-243) -------------------------
    "eva_2d_rpe_tiny": dict(
        attn="eva", x_shape=(1, 14, 14, 128), mask=None,
        args=dict(dim=128, num_heads=2, window_size=7, attn_2d=True, use_rpe=True,
                  num_landmarks=49, adaptive_proj="default")),
    "eva_2d_cfg1": dict(  # BASELINE.json configs[0]: [B,H,N,d]=[2,8,196,64], 49 landmarks
        attn="eva", x_shape=(2, 14, 14, 512), mask=None,
        args=dict(dim=512, num_heads=8, window_size=7, attn_2d=True, use_rpe=True,
                  num_landmarks=49, adaptive_proj="default")),
    "eva_2d_784": dict(  # DeiT-tiny-p8 geometry
        attn="eva", x_shape=(1, 28, 28, 192), mask=None,
        args=dict(dim=192, num_heads=3, window_size=7, attn_2d=True, use_rpe=True,
                  num_landmarks=49, adaptive_proj="default")),
    "eva_2d_overlap_rpe": dict(
        attn="eva", x_shape=(1, 14, 14, 128), mask=None,
        args=dict(dim=128, num_heads=2, window_size=7, attn_2d=True, use_rpe=True,
                  overlap_window=True, num_landmarks=49, adaptive_proj="default")),
    "eva_2d_pvt_w8": dict(  # PvT-style: window 8, 36 landmarks (SURVEY 7 caveat), d=32
        attn="eva", x_shape=(1, 24, 24, 64), mask=None,
        args=dict(dim=64, num_heads=2, window_size=8, attn_2d=True, use_rpe=False,
                  num_landmarks=36, adaptive_proj="default")),
    "eva_2d_t5": dict(
        attn="eva", x_shape=(1, 14, 14, 128), mask=None,
        args=dict(dim=128, num_heads=2, window_size=7, attn_2d=True, use_t5_rpe=True,
                  num_landmarks=49, adaptive_proj="no-ln")),
    "eva_2d_L100": dict(  # 100 landmarks (> the 64 the 16-bit window kernels hold): the generic fp32 kernels take it (round 5)
        attn="eva", x_shape=(1, 20, 20, 64), mask=None,
        args=dict(dim=64, num_heads=2, window_size=5, attn_2d=True, use_rpe=True,
                  num_landmarks=100, adaptive_proj="default")),
    "eva_1d_mask_overlap_t5": dict(  # N=50 not a multiple of w=8 -> padded to 56, 7 chunks of 8
        attn="eva", x_shape=(2, 50, 128), mask=("tail", [0, 9]),
        args=dict(dim=128, num_heads=2, window_size=8, attn_2d=False, use_t5_rpe=True,
                  overlap_window=True, num_landmarks=7, adaptive_proj="default")),
    "eva_1d_nomask_none": dict(
        attn="eva", x_shape=(2, 64, 128), mask=None,
        args=dict(dim=128, num_heads=2, window_size=16, attn_2d=False, use_rpe=True,
                  num_landmarks=8, adaptive_proj="none")),
    # ---------------- LARA (efficient_attention/lara.py:14-267) -----------------------
    "lara_2d_poolmixed_tiny": dict(
        attn="lara", x_shape=(1, 14, 14, 128), mask=None,
        args=dict(dim=128, num_heads=2, num_landmarks=49, proposal_gen="pool-mixed",
                  mis_type="mis-opt", alpha_coeff=2.0)),
    "lara_2d_poolmixed_784": dict(  # BASELINE.json configs[2] geometry (headline metric)
        attn="lara", x_shape=(1, 28, 28, 192), mask=None,
        args=dict(dim=192, num_heads=3, num_landmarks=49, proposal_gen="pool-mixed",
                  mis_type="mis-opt", alpha_coeff=2.0)),
    "lara_2d_L144": dict(  # 144 samples (> the 128 of the 16-bit estimator kernels): the generic fp32 kernels take it (round 5)
        attn="lara", x_shape=(1, 24, 24, 64), mask=None,
        args=dict(dim=64, num_heads=2, num_landmarks=144, proposal_gen="pool-mixed",
                  mis_type="mis-opt", alpha_coeff=2.0)),
    "lara_2d_pool_biased": dict(
        attn="lara", x_shape=(1, 14, 14, 128), mask=None,
        args=dict(dim=128, num_heads=2, num_landmarks=49, proposal_gen="pool",
                  mis_type="mis-biased")),
    "lara_2d_vmixed_bh": dict(
        attn="lara", x_shape=(1, 14, 14, 128), mask=None,
        args=dict(dim=128, num_heads=2, num_landmarks=16, proposal_gen="pool-vmixed",
                  mis_type="mis-bh")),
    "lara_2d_dense_antithetic": dict(
        attn="lara", x_shape=(1, 14, 14, 128), mask=None,
        args=dict(dim=128, num_heads=2, num_landmarks=16, proposal_gen="pool-mixed",
                  pool_module_type="dense", use_antithetics=True, mis_type="mis-opt")),
    "lara_2d_noparam_multisample": dict(
        attn="lara", x_shape=(1, 14, 14, 128), mask=None,
        args=dict(dim=128, num_heads=2, num_landmarks=16, proposal_gen="no-param-pool",
                  use_multisample=True, mis_type="mis-opt")),
    "lara_1d_uneven_mask": dict(  # 50 tokens / 7 landmarks -> uneven-split branch, pad mask
        attn="lara", x_shape=(2, 50, 128), mask=("tail", [0, 9]),
        args=dict(dim=128, num_heads=2, num_landmarks=7, proposal_gen="adaptive-1d",
                  mis_type="mis-opt")),
    "lara_1d_even": dict(
        attn="lara", x_shape=(2, 64, 128), mask=None,
        args=dict(dim=128, num_heads=2, num_landmarks=16, proposal_gen="adaptive-1d",
                  mis_type="mis-opt", alpha_coeff=1.0)),
    # ---------------- softmax baseline (abstract_attention.py:41-140) ------------------
    "softmax_1d_mask": dict(
        attn="softmax", x_shape=(2, 50, 128), mask=("tail", [0, 9]),
        args=dict(dim=128, num_heads=2)),
    "softmax_1d_dropout": dict(  # attn_drop on the [N,N] probabilities (abstract_attention.py:131)
        attn="softmax", x_shape=(2, 70, 128), mask=("tail", [0, 6]),
        args=dict(dim=128, num_heads=2, attn_drop=0.2)),
    "softmax_2d": dict(
        attn="softmax", x_shape=(1, 14, 14, 128), mask=None,
        args=dict(dim=128, num_heads=2)),
    # ---------------- randomized attention (randomized_attention.py:10-63) -------------
    "ra_exact_2d": dict(  # num_samples = -1: mu = q + softmax(s q k^T) k
        attn="ra", x_shape=(1, 14, 14, 128), mask=None, args=dict(dim=128, num_heads=2, num_samples=-1)),
    "ra_mean_1d": dict(  # num_samples = 0: mu = q + mean(k); the pad mask is ignored by the reference
        attn="ra", x_shape=(2, 50, 128), mask=("tail", [0, 9]), args=dict(dim=128, num_heads=2, num_samples=0)),
    "ra_sampled_1d": dict(  # num_samples = 1: one key index per query (injected draws)
        attn="ra", x_shape=(2, 70, 128), mask=None, args=dict(dim=128, num_heads=2, num_samples=1)),
    # ---------------- ScatterBrain (scatterbrain_attention.py:46-180) ------------------------
    "scatterbrain_1d_mask": dict(  # N = 50 -> padded to 56 (7 windows of 8), pad mask, 1-D rpe table
        attn="scatterbrain", x_shape=(2, 50, 128), mask=("tail", [0, 9]),
        args=dict(dim=128, num_heads=2, window_size=8, attn_2d=False, use_rpe=True, approx_attn_dim=32)),
    "scatterbrain_2d": dict(
        attn="scatterbrain", x_shape=(1, 14, 14, 128), mask=None,
        args=dict(dim=128, num_heads=2, window_size=7, attn_2d=True, use_rpe=True, approx_attn_dim=64)),
    # window overlap: the key side of a window is the extended patch, its out-of-range slots zero padding (phi = 1)
    # The reference returns NaN here whenever a border window's padding slots (phi = 1 each) outweigh the features of
    # the keys outside it, log(sum_all - sum_window + 1e-5) of a negative number: any masked / padded 1-D sequence, a
    # grid of 2 x 2 windows, or simply keys of ordinary size (phi(k) << 1).  The two cases use small inputs (x_scale),
    # for which it is finite.
    "scatterbrain_1d_overlap": dict(
        attn="scatterbrain", x_shape=(2, 128, 128), mask=None, x_scale=0.05,
        args=dict(dim=128, num_heads=2, window_size=8, attn_2d=False, use_rpe=True, approx_attn_dim=32,
                  overlap_window=True)),
    "scatterbrain_2d_overlap": dict(
        attn="scatterbrain", x_shape=(1, 28, 28, 128), mask=None, x_scale=0.05,
        args=dict(dim=128, num_heads=2, window_size=4, attn_2d=True, use_rpe=True, approx_attn_dim=16,
                  overlap_window=True)),
    # ---------------- local baseline (local_attention.py:25-194) -----------------------
    "local_2d_rpe": dict(
        attn="local", x_shape=(1, 14, 14, 128), mask=None,
        args=dict(dim=128, num_heads=2, window_size=7, attn_2d=True, use_rpe=True)),
    "local_1d_overlap_mask": dict(
        attn="local", x_shape=(2, 50, 128), mask=("tail", [0, 9]),
        args=dict(dim=128, num_heads=2, window_size=8, attn_2d=False, use_rpe=True,
                  overlap_window=True)),
    # ---------------- Performer baseline (kernelized_attention.py:223-359) -------------
    "performer_1d_mask": dict(
        attn="performer", x_shape=(2, 50, 128), mask=("tail", [0, 9]),
        args=dict(dim=128, num_heads=2, approx_attn_dim=64, proj_method="favorp")),
    "performer_2d_d32": dict(  # head_dim 32, as the pvt_*2 variants have (vit/models/pvt_legacy.py:419-430)
        attn="performer", x_shape=(1, 14, 14, 128), mask=None,
        args=dict(dim=128, num_heads=4, approx_attn_dim=64, proj_method="favorp"), x_scale=0.25),
    "performer_2d": dict(
        attn="performer", x_shape=(1, 14, 14, 128), mask=None,
        args=dict(dim=128, num_heads=2, approx_attn_dim=64, proj_method="favorp")),
    # Performer and the clamp of its normaliser (kernelized_attention.py:55 `clamp(min=1e-2)`).  With the fixture parameters
    # the two cases above sit ENTIRELY under the clamp (every query's denominator < 1e-2: out = numerator / 1e-2, no gradient
    # through the denominator).  These two pin the other regimes on reference vectors: x_scale 0.55 ->
    # about half of the queries clamped (the kink inside the batch), x_scale 0.3 -> none clamped (the full quotient rule).
    "performer_2d_clamp": dict(
        attn="performer", x_shape=(1, 14, 14, 128), mask=None, x_scale=0.55,
        args=dict(dim=128, num_heads=2, approx_attn_dim=64, proj_method="favorp")),
    "performer_2d_unclamped": dict(
        attn="performer", x_shape=(1, 14, 14, 128), mask=None, x_scale=0.3,
        args=dict(dim=128, num_heads=2, approx_attn_dim=64, proj_method="favorp")),
    # ---------------- a FULLY padded batch row (local / EVA fill -5e4 and stay finite; Performer zeroes the
    # padded features; the reference's softmax and LARA fill -inf and return NaN for such a row -- tests/test_gpu_padding.py)
    "local_1d_fullpad": dict(
        attn="local", x_shape=(2, 24, 128), mask=("tail", [0, 24]),
        args=dict(dim=128, num_heads=2, window_size=8, attn_2d=False, use_rpe=True)),
    "eva_1d_fullpad": dict(
        attn="eva", x_shape=(2, 24, 128), mask=("tail", [5, 24]),
        args=dict(dim=128, num_heads=2, window_size=8, attn_2d=False, use_rpe=True,
                  num_landmarks=3, adaptive_proj="default")),
    "performer_1d_fullpad": dict(
        attn="performer", x_shape=(2, 24, 128), mask=("tail", [0, 24]),
        args=dict(dim=128, num_heads=2, approx_attn_dim=64, proj_method="favorp")),
    # ---------------- causal EVA, training/evaluation path (causal_eva.py:666-790) -----
    # x is batch-first here; generator and tests transpose to the module's time-first layout
    "causal_eva_lm_small": dict(  # the wikitext-103 recipe (README.md:184) scaled down
        attn="causal_eva", x_shape=(2, 64, 128), mask=None,
        args=dict(embed_dim=128, num_heads=2, self_attention=True,
                  attn_args=dict(window_size=16, chunk_size=4, causal=True, adaptive_proj="qk",
                                 use_t5_rpe=True, num_chunks=None, overlap_window=False))),
    "causal_eva_overlap_mask": dict(  # T=50 -> padded to 56; left extension e = w; pad mask
        attn="causal_eva", x_shape=(2, 50, 128), mask=("tail", [0, 9]),
        args=dict(embed_dim=128, num_heads=2, self_attention=True,
                  attn_args=dict(window_size=8, chunk_size=4, causal=True, adaptive_proj="no-ln",
                                 use_t5_rpe=True, num_chunks=None, overlap_window=True))),
    "causal_eva_noncausal_chunks": dict(  # causal flag off, chunk length from num_chunks
        attn="causal_eva", x_shape=(2, 64, 128), mask=("tail", [5, 0]),
        args=dict(embed_dim=128, num_heads=2, self_attention=True,
                  attn_args=dict(window_size=8, chunk_size=None, causal=False, adaptive_proj="qk",
                                 use_t5_rpe=False, num_chunks=8, overlap_window=True))),
    "causal_eva_dropout": dict(  # attention dropout 0.1 (transformer_lm_wiki103's default), pad mask
        attn="causal_eva", x_shape=(2, 48, 128), mask=("tail", [0, 5]),
        args=dict(embed_dim=128, num_heads=2, dropout=0.1, self_attention=True,
                  attn_args=dict(window_size=16, chunk_size=4, causal=True, adaptive_proj="qk",
                                 use_t5_rpe=True, num_chunks=None, overlap_window=True))),
    "causal_eva_qnoise": dict(  # quantization noise on the four projections (causal_eva.py:118-213, 339-351): blocks of 8
        attn="causal_eva", x_shape=(2, 48, 128), mask=("tail", [3, 0]),      # input features dropped with p = 0.25 in training
        args=dict(embed_dim=128, num_heads=2, self_attention=True, q_noise=0.25, qn_block_size=8,
                  attn_args=dict(window_size=16, chunk_size=4, causal=True, adaptive_proj="qk",
                                 use_t5_rpe=True, num_chunks=None, overlap_window=False))),
}

MODES = ("eval", "train")


def ctor_args(case):
    """Constructor kwargs of a case as the factory expects them (causal_eva reads its flags from
    an argparse namespace, causal_eva.py:354-376)."""
    args = dict(case["args"])
    if case["attn"] == "causal_eva":
        args["attn_args"] = argparse.Namespace(**args["attn_args"])
    return args


def call_module(case, mod, x, mask):
    """y = module(x[, mask]) with batch-first x for every variant (causal_eva is a time-first
    (query, key, value) module returning (out, None), causal_eva.py:443-470)."""
    if case["attn"] == "causal_eva":
        xt = x.transpose(0, 1)
        return mod(xt, xt, xt, key_padding_mask=mask)[0].transpose(0, 1)
    return mod(x, mask) if mask is not None else mod(x)


def _seed(name, salt):
    return zlib.crc32(("%s/%s" % (name, salt)).encode()) & 0x7FFFFFFF


def rng_for(name, salt):
    return np.random.Generator(np.random.PCG64(_seed(name, salt)))


def make_params(name, key_shapes):
    """Parameters/buffers for a case from the (key -> shape) table stored in its fixture.

    Values are chosen so that every term matters in a parity check: Linear weights are
    O(1/sqrt(fan_in)), biases are non-zero, LayerNorm affine is not the identity and the
    relative-position tables are O(0.5).  `relative_position_index` is NOT generated here:
    it is an integer buffer the module builds itself (and the fixture stores the
    reference's copy so the two can be compared)."""
    rng = rng_for(name, "params")
    out = {}
    for key in sorted(key_shapes):
        shape = tuple(key_shapes[key])
        if key == "relative_position_index":
            continue
        base = rng.standard_normal(shape).astype(np.float32)
        leaf = key.split(".")[-1]
        if key.endswith("relative_attention_bias.weight") or "bias_table" in key:
            val = 0.5 * base
        elif key in ("eval_proj", "random_proj"):
            val = base
        elif leaf == "weight" and len(shape) == 2:
            val = base / np.sqrt(shape[1])
        elif leaf == "weight" and len(shape) == 1:      # LayerNorm gain
            val = 1.0 + 0.2 * base
        elif leaf == "bias":
            val = 0.1 * base
        else:
            val = 0.1 * base
        out[key] = val.astype(np.float32)
    return out


def make_inputs(name):
    case = CASES[name]
    x = (case.get("x_scale", 1.0) * rng_for(name, "x").standard_normal(case["x_shape"])).astype(np.float32)
    g = rng_for(name, "g").standard_normal(case["x_shape"]).astype(np.float32)
    mask = None
    if case["mask"] is not None:
        kind, pads = case["mask"]
        assert kind == "tail"
        B, N = case["x_shape"][0], case["x_shape"][1]
        mask = np.zeros((B, N), dtype=bool)
        for b, p in enumerate(pads):
            if p:
                mask[b, N - p:] = True
    return x, g, mask


def make_noise(name, shape, call_idx=0):
    """Standard-normal sampling noise for training mode: the i-th randn/randn_like call
    inside one forward gets stream (name, 'noise<i>')."""
    return rng_for(name, "noise%d" % call_idx).standard_normal(tuple(shape)).astype(np.float32)


def make_index(name, shape, high, call_idx=0):
    """Key indices standing in for torch.multinomial draws (any index has positive probability)."""
    return rng_for(name, "index%d" % call_idx).integers(0, high, size=tuple(shape)).astype(np.int64)


def make_keep(name, shape, p_drop, call_idx=0):
    """0/1 keep decisions of an attention-dropout call (probability p_drop of dropping)."""
    u = rng_for(name, "keep%d" % call_idx).random(tuple(shape), dtype=np.float32)
    return (u >= p_drop).astype(np.float32)


def make_block_mask(name, n, p, call_idx=0):
    """0/1 drop decisions of one quantization-noise draw (`mask.bernoulli_(p)` over the n weight blocks of a projection,
    causal_eva.py:175-179): 1 = the block is zeroed."""
    u = rng_for(name, "qnoise%d" % call_idx).random((int(n),), dtype=np.float32)
    return (u < p).astype(np.float32)


GRAD_FULL_MAX = 16384     # parameter grads larger than this are stored as a subsample
GRAD_SAMPLES = 4096


def grad_sample_index(name, key, numel):
    """Flat indices at which a large parameter gradient is stored in the fixture."""
    return rng_for(name, "gradidx/" + key).integers(0, numel, size=GRAD_SAMPLES)


def pack_grad(name, key, arr):
    """-> dict of arrays to store for one parameter gradient (full, or subsample+moments)."""
    arr = np.asarray(arr, np.float32)
    if arr.size <= GRAD_FULL_MAX:
        return {"": arr}
    idx = grad_sample_index(name, key, arr.size)
    flat = arr.reshape(-1)
    return {".sample": flat[idx],
            ".moments": np.array([flat.sum(dtype=np.float64),
                                  np.abs(flat).sum(dtype=np.float64),
                                  (flat.astype(np.float64) ** 2).sum()], np.float64)}
