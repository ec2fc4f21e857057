#!/usr/bin/env python3
"""bench.py -- attention-layer forward+backward throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 30 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one synthetic batch: x -> attention layer (qkv
projection, HIP attention core, output projection) -> loss -> backward -> SGD step, under bf16
autocast, one process per GPU; with N > 1 the layer is wrapped in DistributedDataParallel (RCCL
all-reduce of the parameter gradients over xGMI, weak scaling: fixed per-GPU batch).  Inputs are
resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

Default workload (config.workload): BASELINE.json configs[2] geometry, the one the metric is
quoted on -- DeiT-tiny-p8 tokens (N = 28x28 = 784, 3 heads, d = 64), per-GPU batch 128.
`--attn eva|lara|softmax|local|performer|ra|scatterbrain` selects the attention (default: see DEFAULT_ATTN);
`--attn causal_eva --workload lm` is the wikitext-103 decoder self-attention.
"""
import argparse
import json
import os
import sys
import tempfile
import time
import warnings

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "efficient-attention_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

DEFAULT_ATTN = "lara"
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 achievable
MFMA_PEAK_TFLOPS = 2500.0      # dense bf16 / fp16 matrix peak (MI355X_MICROARCH.md; the headline figure with sparsity is 2x)
BYTES_PER_TOKEN_HEAD = 1536    # fwd (q,k,v,out) + bwd (q,k,v,out,dout,dq,dk,dv) at d=64, bf16


OVERRIDES = {}      # --window / --landmarks


def _emit(line):
    """The JSON line goes out last: RCCL's banner sits in C stdio buffers until they are flushed."""
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    print(json.dumps(line), flush=True)


def attn_args(attn, dim, heads, seq):
    args = _attn_args(attn, dim, heads, seq)
    for k, v in OVERRIDES.items():
        if k in args and v is not None:
            args[k] = v
    return args


def _attn_args(attn, dim, heads, seq):
    """seq: (side, side) for the vit recipes (cfg2 / cfg3), (N,) for the fairseq-style 1-D ones (cfg5)."""
    base = dict(dim=dim, num_heads=heads, qkv_bias=True, attn_drop=0.0, proj_drop=0.0)
    seq = _seq(seq)
    two_d = len(seq) == 2
    if attn == "eva":
        if two_d:
            base.update(window_size=7, attn_2d=True, use_rpe=True, num_landmarks=49, adaptive_proj="default")
        else:
            base.update(window_size=16, attn_2d=False, use_t5_rpe=True, overlap_window=True, num_landmarks=8,
                        adaptive_proj="default")
    elif attn == "local":
        base.update(window_size=7 if two_d else 16, attn_2d=two_d, use_rpe=True)
    elif attn == "scatterbrain":
        base.update(window_size=7 if two_d else 16, attn_2d=two_d, use_rpe=True, approx_attn_dim=64)
    elif attn == "lara":
        if two_d:
            base.update(num_landmarks=49, proposal_gen="pool-mixed", mis_type="mis-opt", alpha_coeff=2.0)
        else:
            base.update(num_landmarks=49, proposal_gen="adaptive-1d", mis_type="mis-opt")
    elif attn == "performer":
        base.update(approx_attn_dim=64, proj_method="favorp")
    return base


def _seq(grid):
    return tuple(grid) if isinstance(grid, (tuple, list)) else (grid, grid)


# causal_eva on the wikitext-103 recipe (reference README.md:184; main.sh:52-81: transformer_lm_wiki103
# = embed 1024 / 8 heads, attention dropout 0.1, --tokens-per-sample 512, --max-tokens 9216 -> 18
# samples per GPU).
LM_ATTENTION_DROPOUT = 0.1
LM_ATTN_ARGS = dict(window_size=128, chunk_size=8, causal=True, adaptive_proj="qk", use_t5_rpe=True,
                    num_chunks=None, overlap_window=False)


def build_layer(attn, dim, heads, grid, device):
    import efficient_attention as ea
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if attn == "causal_eva":
            aa = dict(LM_ATTN_ARGS)
            if OVERRIDES.get("window_size"):
                aa["window_size"] = OVERRIDES["window_size"]
            return ea.AttentionFactory.build_attention(attn, dict(
                embed_dim=dim, num_heads=heads, dropout=LM_ATTENTION_DROPOUT, self_attention=True,
                attn_args=argparse.Namespace(**aa))).to(device)
        return ea.AttentionFactory.build_attention(attn, attn_args(attn, dim, heads, _seq(grid))).to(device)


def cpu_baseline(attn, dim, heads, grid, batch=128, budget_s=30.0):
    """The oracle (CPU restatement of the reference, kind='port') timed on this host's cores the way SURVEY.md 8(d) prescribes:
    same layer, token geometry and -- when it fits the time budget -- the GPU line's batch, fp32, forward + backward; per
    thread count 3 warm-up + >= 10 timed iterations, MEDIAN; all host cores and 8 threads, both reported (`threads`), the
    faster one is `value`.  The batch is halved until 2 x 13 iterations are estimated to fit `budget_s` (the estimate is one
    probe step at batch 8); `sample` states the batch that ran."""
    import oracle
    torch.manual_seed(1234)
    layer = build_layer(attn, dim, heads, grid, "cpu")
    params = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in layer.state_dict().items()}
    seq = _seq(grid)
    if attn == "causal_eva":
        args = dict(embed_dim=dim, num_heads=heads, dropout=LM_ATTENTION_DROPOUT, attn_args=dict(
            LM_ATTN_ARGS, window_size=OVERRIDES.get("window_size") or LM_ATTN_ARGS["window_size"]))
    else:
        args = attn_args(attn, dim, heads, seq)
    noise_fn = lambda shape: torch.randn(*shape)  # noqa: E731
    ntok = 1
    for v in seq:
        ntok *= v

    def make_step(B):
        x = torch.randn(B, *seq, dim, requires_grad=True)
        g = torch.randn(B, *seq, dim)

        def step():
            for p in params.values():
                p.grad = None
            x.grad = None
            y = oracle.module_forward(attn, args, params, x, None, training=True, noise_fn=noise_fn,
                                      keep_fn=lambda shape: (torch.rand(*shape) >= LM_ATTENTION_DROPOUT).float(),
                                      index_fn=lambda shape: torch.randint(0, shape[-1], tuple(shape)))
            (y * g).sum().backward()
        return step

    def timed_once(step):
        t0 = time.perf_counter()
        step()
        return time.perf_counter() - t0

    WARM, TIMED = 3, 10
    all_cores = torch.get_num_threads()
    counts = sorted({all_cores, min(8, all_cores)})
    torch.set_num_threads(counts[0])
    probe = make_step(8)
    probe()
    per_elem = timed_once(probe) / 8                      # seconds per batch element at the small thread count
    B = max(1, int(batch))
    while B > 8 and per_elem * B * (WARM + TIMED) * len(counts) > budget_s:
        B //= 2
    step = make_step(B)
    per_threads = {}
    for threads in counts:
        torch.set_num_threads(threads)
        for _ in range(WARM):
            step()
        ts = sorted(timed_once(step) for _ in range(TIMED))
        med = 0.5 * (ts[(TIMED - 1) // 2] + ts[TIMED // 2])
        per_threads[threads] = B * ntok / med
    torch.set_num_threads(all_cores)
    threads, tok_s = max(per_threads.items(), key=lambda kv: kv[1])
    return {"value": tok_s, "unit": "tokens/s", "cores": threads, "kind": "port",
            "threads": {str(t): round(v, 1) for t, v in per_threads.items()},
            "batch": B, "warmup": WARM, "timed": TIMED, "statistic": "median",
            "sample": "oracle layer (%s, fp32) fwd+bwd at batch %d (GPU line: batch %d), N=%d, dim %d; per thread count %d warm-up + "
                      "%d timed iterations, median [%s]"
                      % (attn, B, batch, ntok, dim, WARM, TIMED,
                         "; ".join("%d threads: %.0f tokens/s" % (t, v) for t, v in sorted(per_threads.items())))}


def _sgd(big, small, lr):
    """Plain SGD, p -= lr * grad (what torch.optim.SGD(lr).step() computes): one element-wise launch per weight matrix
    (>= 16384 elements) and ONE multi-tensor launch for the small vectors.  The multi-tensor kernel walks 65536-element chunks
    per workgroup: it would run the two matrices of a 192-wide layer on ~3 workgroups (17 us against 2 x 4.6), and for the LM
    layer's six large matrices (66 chunks) it measured 8 us SLOWER per step than six separate launches (A/B on one box,
    profiles/r06_lm_stacked_sgd_ab.txt: 0.9616 against 0.9525 ms) -- so the rule is the same at every width."""
    for prm in big:
        if prm.grad is not None:
            prm.data.add_(prm.grad, alpha=-lr)
    ps = [prm for prm in small if prm.grad is not None]
    if ps:
        torch._foreach_add_([prm.data for prm in ps], [prm.grad for prm in ps], alpha=-lr)


def measure_workload(attn, B, C, H, seq, dev, steps=10, warmup=3, tune=True, overrides=None):
    """One more workload of BASELINE.json next to the headline one (N = 196 / 4096): the same layer step -- fwd + bwd + SGD
    under bf16 autocast, captured in a hipGraph -- timed over `steps` replays.  Returns tokens/s and the layer-level
    fraction of the HBM roofline (1536*h bytes per token at d = 64).  overrides: attention arguments that
    replace the recipe's for this layer only (the PvT stages: window_size 8, 36 landmarks)."""
    N = 1
    for s_ in seq:
        N *= s_
    d = C // H
    keep = dict(OVERRIDES)
    OVERRIDES.update(overrides or {})
    try:
        layer = build_layer(attn, C, H, seq, dev)
    finally:
        OVERRIDES.clear()
        OVERRIDES.update(keep)
    layer.train()
    lm = attn == "causal_eva"                                        # fairseq decoder self-attention: time-first, (q, k, v) call
    xshape = (seq[0], B, C) if lm else (B,) + tuple(seq) + (C,)
    x = torch.randn(*xshape, device=dev, requires_grad=True)
    g = torch.randn(*xshape, device=dev).to(torch.bfloat16)
    params = list(layer.parameters())
    big = [prm for prm in params if prm.numel() >= 16384]
    small = [prm for prm in params if prm.numel() < 16384]

    def step():
        for prm in params:
            prm.grad = None
        x.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = layer(x, x, x)[0] if lm else layer(x)
        y.backward(g)
        _sgd(big, small, 1e-3)
    if tune:
        import torch.cuda.tunable as tunable
        tunable.tuning_enable(True)
    for _ in range(max(warmup, 1)):
        step()
    torch.cuda.synchronize()
    if tune:
        tunable.tuning_enable(False)
    graphed = True
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            step()
        torch.cuda.current_stream().wait_stream(s)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            step()
        run = gr.replay
    except Exception:
        torch.cuda.synchronize()
        run, graphed = step, False
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    # clock pre-warm as in main() (untimed), then three timed blocks of `steps` steps: the median block is reported
    pw = float(os.environ.get("EA_BENCH_PREWARM_MS", "50")) * 0.4
    tpw = time.perf_counter()
    while (time.perf_counter() - tpw) * 1e3 < pw:
        for _ in range(4):
            run()
        torch.cuda.synchronize()
    els = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        torch.cuda.synchronize()
        els.append(time.perf_counter() - t0)
    el = sorted(els)[1]
    tok_s = B * N * steps / el
    return {"x": [B] + list(seq) + [C], "seq_len": N, "heads": H, "head_dim": d, "ms_per_step": round(el / steps * 1e3, 4),
            "tokens_per_s": tok_s, "hipgraph": graphed,
            "layer_hbm_roofline_frac": round(tok_s * BYTES_PER_TOKEN_HEAD * H * d / 64 / (HBM_PEAK_GBS * 1e9), 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--attn", default=DEFAULT_ATTN)
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch")
    ap.add_argument("--grid", type=int, default=28, help="token grid side (28 -> N=784)")
    ap.add_argument("--dim", type=int, default=192)
    ap.add_argument("--heads", type=int, default=3)
    ap.add_argument("--window", type=int, default=None, help="override window_size (eva / local), e.g. 8 for the PvT stages")
    ap.add_argument("--landmarks", type=int, default=None, help="override num_landmarks (eva / lara), e.g. 36 for the PvT stages")
    ap.add_argument("--workload", default="cfg3", choices=["cfg3", "cfg2", "cfg5", "lm", "model_cfg2", "model_cfg3",
                                                           "model_cfg4", "model_cfg5"],
                    help="cfg3 (default, the metric's config): [128,28,28,192] h=3; cfg2: [128,14,14,192] h=3; "
                         "cfg5: 1-D [16,4096,512] h=8 (BASELINE.json configs / SURVEY.md 8d); lm: the wikitext-103 "
                         "decoder self-attention, [18,512,1024] h=8 (use with --attn causal_eva); model_cfg2..5: the WHOLE "
                         "models of BASELINE.json configs 2-5 (ea_harness: DeiT-tiny-p16 EVA, DeiT-tiny-p8 LARA, PvTv2-b2 "
                         "EVA, wmt_en_de encoder LARA) as a synthetic training step, eager and captured")
    ap.add_argument("--no-graph", action="store_true", help="do not capture the step in a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the cfg2 / cfg5 lines of `other_workloads`")
    ap.add_argument("--no-gemm-tune", action="store_true",
                    help="skip PyTorch TunableOp selection of the hipBLASLt/rocBLAS projection GEMMs")
    ap.add_argument("--gemm-tune-file", default=None,
                    help="TunableOp results file: read (no tuning) if it exists, else tuned and written at exit")
    a = ap.parse_args()
    if a.workload.startswith("model_"):
        return main_model(a)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # EA_BENCH_BACKEND=gloo + EA_BENCH_ONE_DEVICE=1: dev-only way to exercise the N > 1 code path
    # (DDP hooks, barriers, max-over-ranks) on a single-GPU box; never used by the driver.
    backend = os.environ.get("EA_BENCH_BACKEND", "nccl")
    # EA_BENCH_FORCE_DDP=1: dev-only, run the N > 1 code path (process group, DDP wrapper, eager
    # stepping) with a single rank to measure its host-side overhead on one GPU
    ddp = world > 1 or bool(os.environ.get("EA_BENCH_FORCE_DDP"))
    if os.environ.get("EA_BENCH_ONE_DEVICE"):
        local = 0
    torch.cuda.set_device(local)                # before the process group: RCCL binds to the current device
    dev = torch.device("cuda", local)
    if ddp:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group(backend, rank=0, world_size=1)
        else:
            dist.init_process_group(backend)
    torch.manual_seed(1234 + rank)

    OVERRIDES.update(window_size=a.window, num_landmarks=a.landmarks)
    B, C, H = a.batch, a.dim, a.heads
    seq = (a.grid, a.grid)
    if a.workload == "cfg2":
        seq = (14, 14)
    elif a.workload == "cfg5":
        B, C, H, seq = (16 if a.batch == 128 else a.batch), 512, 8, (4096,)
    elif a.workload == "lm":
        B, C, H, seq = (18 if a.batch == 128 else a.batch), 1024, 8, (512,)
    if (a.attn == "causal_eva") != (a.workload == "lm"):
        ap.error("--attn causal_eva goes with --workload lm (a time-first 1-D decoder self-attention)")
    G = seq
    N = 1
    for s_ in seq:
        N *= s_
    d = C // H
    layer = build_layer(a.attn, C, H, G, dev)
    layer.train()
    model = layer
    LR = 1e-3
    xshape = (seq[0], B, C) if a.attn == "causal_eva" else (B,) + tuple(seq) + (C,)   # fairseq is time-first
    x = torch.randn(*xshape, device=dev, requires_grad=True)
    g = torch.randn(*xshape, device=dev).to(torch.bfloat16)        # cotangent of y, in y's dtype
    if a.attn == "causal_eva":
        def model(t):
            return layer(t, t, t)[0]

    from efficient_attention import _ops

    # Which projections are library GEMMs depends on the width: a 64..256-wide layer runs every product on this library's
    # own kernels (ea_linear*, ea_wgrad*, ea_linear_dgrad*) and has NO library GEMM in its step; wider layers (cfg5, the LM
    # workload, PvT stages 3-4) take the kernels of ea_gemm.hip for widths it covers and hipBLASLt / rocBLAS otherwise.  For
    # the latter the untimed warm-up steps run with PyTorch's TunableOp selecting the solution per shape (the default
    # heuristic picks poor tiles for skinny shapes); selection is frozen before the step is captured / timed.
    # `config.gemm_tunableop` says whether a library GEMM actually ran under it (TunableOp recorded a result).
    tune = not a.no_gemm_tune
    if tune:
        import torch.cuda.tunable as tunable
        tunable.enable(True)
        if a.gemm_tune_file:
            tunable.set_filename(a.gemm_tune_file)
            tunable.tuning_enable(not os.path.exists(a.gemm_tune_file))
        else:
            tunable.set_filename(os.path.join(tempfile.gettempdir(), "ea_bench_tunableop_%d.csv" % os.getpid()))
            tunable.tuning_enable(True)

    def fwd_bwd():
        for prm in params:
            prm.grad = None
        x.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = model(x)
        y.backward(g)                       # == (y * g).sum().backward() without the loss arithmetic

    params = [prm for prm in layer.parameters()]
    if not ddp:
        # Plain SGD, p -= lr * grad (what torch.optim.SGD(lr=LR).step() computes), issued as: one element-wise update for
        # each weight matrix and ONE multi-tensor update for the ten small vectors.  The all-in-one multi-tensor kernel
        # walks 65536-element chunks per workgroup, i.e. runs the two matrices on ~3 workgroups (17 us); twelve separate
        # updates are twelve launches.
        big = [prm for prm in params if prm.numel() >= 16384]
        small = [prm for prm in params if prm.numel() < 16384]

        def sgd_step():
            _sgd(big, small, LR)

        def step():
            fwd_bwd()
            sgd_step()
        parts = [step]
    else:
        # Data parallel over the batch (the path has no other exchange): every rank's parameter
        # gradients are packed into ONE flat fp32 bucket (0.16 M values), all-reduced over RCCL, and
        # applied as plain SGD with the 1/world averaging folded into the step size -- what
        # DistributedDataParallel + optim.SGD compute, without DDP's per-iteration host work
        # (measured: the DDP wrapper makes this 0.9 ms step host-bound at 1.44 ms).
        from efficient_attention.data_parallel import FlatGradBucket
        bucket = FlatGradBucket(params, dev)
        bucket.broadcast_parameters(0)

        # Two equivalent schedules of a step; which one runs is MEASURED below, on every launch:
        #   pipelined : [update + forward/backward + pack] as ONE captured hipGraph, then the all-reduce
        #               (the update applies the previous step's summed gradients -- zeros before the
        #               first step -- so every step still holds one update, one forward/backward and
        #               one all-reduce: two host calls and one stream hand-over per step);
        #   three_part: [forward/backward + pack] | all-reduce | [update]  (three calls, two hand-overs).
        # On a single-rank RCCL group the pipelined form costs 0.83 ms against 0.89 ms; with two
        # ranks time-slicing ONE device over gloo (the dev configuration) it was 100x slower, so the
        # choice is not hard-wired.
        from efficient_attention.data_parallel import ddp_schedules, select_schedule
        schemes = ddp_schedules(fwd_bwd, bucket, LR)
        forced = os.environ.get("EA_BENCH_DDP_SCHEME")
        if forced:
            schemes = {forced: schemes[forced]}
        parts = schemes.get("three_part", list(schemes.values())[0])

    def step():
        for f in parts:
            f()

    for _ in range(max(a.warmup, 1)):
        step()
    torch.cuda.synchronize()
    gemm_tuned = False
    if tune:
        tunable.tuning_enable(False)
        try:                                   # did a library GEMM run at all?  (192-wide steps have none)
            gemm_tuned = len(tunable.get_results()) > 0
        except Exception:
            gemm_tuned = None

    def capture(fn):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn()
        torch.cuda.current_stream().wait_stream(s)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            fn()
        return gr

    def prepare(fns):
        """-> (callables of one step, captured?)"""
        if a.no_graph:
            return fns, False
        try:
            out = [f if f.__name__ == "reduce" else capture(f).replay for f in fns]
            for f in out:
                f()
            torch.cuda.synchronize()
            return out, True
        except Exception as ex:  # capture is an optimisation, never a requirement
            if rank == 0:
                print("graph capture unavailable (%s); timing eager" % str(ex).split("\n")[0], file=sys.stderr)
            torch.cuda.synchronize()
            return fns, False

    ddp_scheme = None
    if not ddp:
        run_parts, graphed = prepare(parts)
    else:
        # time a few steps of every schedule (max over ranks, so all ranks take the same decision): the selection itself is
        # efficient_attention.data_parallel.select_schedule, unit-tested on CPU with a fake clock and with two gloo ranks
        def _reduce_max(sec):
            tsel = torch.tensor([sec], dtype=torch.float64, device=dev)
            dist.all_reduce(tsel, op=dist.ReduceOp.MAX)
            return float(tsel.item())
        ddp_scheme, run_parts, graphed, _ = select_schedule(
            schemes, prepare, torch.cuda.synchronize, _reduce_max, barrier=dist.barrier if world > 1 else None)

    def run():
        for f in run_parts:
            f()
    # Clock pre-warm (round 6; untimed, NOT part of the W warm-up steps or of the K timed ones): the timed region is K x ~0.5 ms
    # = ~10 ms at the driver's K = 20, which is as long as the GPU's power management takes to reach its sustained clocks
    # after the host-bound capture phase -- rounds 1-5 reported a first block 1-2 % slower than the five blocks that follow it
    # (`ms_per_step` 0.507 vs `ms_per_step_blocks.median` 0.497 on one box).  The same captured step is replayed for
    # EA_BENCH_PREWARM_MS (default 50) milliseconds of wall time first; `prewarm_ms` in the JSON line says so.
    prewarm_ms = float(os.environ.get("EA_BENCH_PREWARM_MS", "50"))
    # (N > 1: a step holds the gradient all-reduce, so the ranks must replay the same number of steps -- rank 0's clock decides)
    def _agree(done):
        flag = torch.tensor([1 if done else 0], dtype=torch.int32, device=dev)
        dist.broadcast(flag, 0)
        return bool(flag.item())
    from efficient_attention.data_parallel import prewarm_replays
    prewarm_replays(run, prewarm_ms, torch.cuda.synchronize, agree=_agree if world > 1 else None)
    for _ in range(a.warmup):
        run()

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    # ---- spread of the measurement: five more blocks of `steps` steps each (not part of `value`) ----
    blocks_ms = []
    if world == 1:
        for _ in range(5):
            torch.cuda.synchronize()
            tb = time.perf_counter()
            for _ in range(a.steps):
                run()
            torch.cuda.synchronize()
            blocks_ms.append((time.perf_counter() - tb) / a.steps * 1e3)

    # ---- the same step run EAGERLY (the reference's call sites run eagerly, vit/engine.py:47-64) ----
    # Host-bound, so host jitter shows: five blocks, the median is reported (one 20-step block read 0.67 .. 0.93 ms on the
    # same box within a minute), min / max next to it.
    for _ in range(10):
        step()
    eager_blocks = []
    n_eager = max(min(a.steps, 40), 1)
    for _ in range(5 if world == 1 else 1):
        torch.cuda.synchronize()
        te = time.perf_counter()
        for _ in range(n_eager):
            step()
        torch.cuda.synchronize()
        eager_blocks.append((time.perf_counter() - te) / n_eager * 1e3)
    eager_ms = sorted(eager_blocks)[len(eager_blocks) // 2]

    # ---- instrumented eager pass: HIP events around every launch of the dominant kernel ----
    _ops.KERNEL_TIMER.enable()
    for _ in range(min(a.steps, 10)):
        step()
    torch.cuda.synchronize()
    ktimes = _ops.KERNEL_TIMER.summary()
    _ops.KERNEL_TIMER.disable()

    # achievable HBM bandwidth of this GPU, babel-stream style: the library's own 16-byte-per-lane copy kernel
    # over 512 MB (read + write counted), next to the nominal 8 TB/s; torch's copy_ alongside for reference
    copy_gbs = copy_torch_gbs = None
    if rank == 0:
        from efficient_attention import _native as nv
        src = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
        dst = torch.empty_like(src)

        def _time_copy(fn):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return 10 * 2 * src.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9
        copy_gbs = _time_copy(lambda: nv.call("ea_stream_copy", nv.ptr(src), nv.ptr(dst), src.numel(), nv.stream()))
        copy_torch_gbs = _time_copy(lambda: dst.copy_(src))
        del src, dst

    roof = None
    step_bytes = None
    if rank == 0:
        tokens = B * N * world * a.steps
        value = tokens / el
        unit_bytes = B * H * N * d * 2                   # one [B,H,N,D] tensor in the I/O dtype
        # committed counter summaries, used only while they describe THIS build of the library
        # (tools/summarize_profile.py / summarize_sq.py stamp the .so's sha256 into them)
        import hashlib
        lib_path = os.path.join(ROOT, "efficient-attention_amd", "lib", "libea_hip.so")
        sha = hashlib.sha256(open(lib_path, "rb").read()).hexdigest()[:16]

        def _stamped(fname):
            path = os.path.join(ROOT, "profiles", fname)
            if not os.path.exists(path):
                return {}
            rec = json.load(open(path))
            return rec if rec.get("_lib_sha256") == sha else {}
        wl_sfx = "_" + a.workload if a.workload in ("cfg2", "cfg5") else ""
        pmc, sq = _stamped("pmc_%s%s.json" % (a.attn, wl_sfx)), _stamped("sq_%s%s.json" % (a.attn, wl_sfx))

        def _entry(name, st, algo_bytes):
            # achieved = SUMMED algorithmic bytes / SUMMED time of every launch under the label (== bytes / avg duration
            # when all launches of the label have one shape, which the per-shape labels guarantee)
            ach = algo_bytes * st["n"] / (st["total_ms"] * 1e-3) / 1e9
            mu = sq.get(name, {}).get("mfma_util") if isinstance(sq.get(name), dict) else None
            # two clocks: `avg_us` / `achieved` / `frac` = HIP events around every launch of an EAGER instrumented pass, live in
            # this run; `rocprof_avg_us` / `frac_rocprof` = the kernel's average duration in the rocprofv3 kernel trace of the
            # CAPTURED step (profiles/pmc_<attn>.json, only while it is stamped with this build of the library).  The event
            # clock includes the record overhead: it read 58.9 us where the trace read 54.0
            rp = (pmc.get("_rocprof_avg_us") or {}).get(name.split("[")[0])
            e = {"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": pmc.get(name), "avg_us": round(st["avg_ms"] * 1e3, 2),
                 "clock": "hip_events_eager",
                 "rocprof_avg_us": rp, "frac_rocprof": None if not rp else round(algo_bytes / (rp * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                 "launches": st["n"], "algo_bytes_per_launch": algo_bytes,
                 "mfma_util": None if mu is None else round(mu, 4)}
            return e
        # `roofline`: the kernel with the largest total time AMONG THOSE SURVEY 8(d) PRICES -- the attention kernels that
        # stream q, k, v, out and their gradients (KERNEL_ALGO_UNITS: [B,H,N,D] tensors read + written per launch).  The
        # projection kernels (the module edges) have no 8(d) bytes: they are reported under
        # `projection_kernels`, each label one shape, priced on the activations that launch has to move.
        priced = [(k, v) for k, v in ktimes.items() if _ops.KERNEL_ALGO_UNITS.get(k)]
        if priced:
            name, st = max(priced, key=lambda kv: kv[1]["total_ms"])
            roof = _entry(name, st, _ops.KERNEL_ALGO_UNITS[name] * unit_bytes)
            roof["attention_kernels"] = {k: _entry(k, v, _ops.KERNEL_ALGO_UNITS[k] * unit_bytes) for k, v in priced}
            for v in roof["attention_kernels"].values():
                del v["kernel"], v["bound"], v["peak"], v["unit"], v["clock"]
            roof["projection_kernels"] = {}
            for k, v in ktimes.items():
                rec = _ops.LABEL_ALGO_BYTES.get(k)
                if rec and rec[1]:
                    e = _entry(k, v, rec[0] // rec[1])
                    del e["kernel"], e["bound"], e["peak"], e["unit"], e["traffic"], e["mfma_util"], e["clock"]
                    roof["projection_kernels"][k] = e
            roof["all_kernels_avg_us"] = {k: round(v["avg_ms"] * 1e3, 2) for k, v in ktimes.items()}
            if name.startswith("ea_softmax_attn") and N >= 600:
                # the softmax baseline is MFMA-bound for N >~ 600.  ALGORITHMIC FLOPs: forward 4 N^2 d per
                # (b,h) (QK^T, PV), backward 10 N^2 d (dV, dP, dQ, dK + the S recompute a flash-style backward cannot
                # avoid) = 3.5 x forward for the pair; executed-but-redundant products are not counted
                fl = (4 if name.endswith("fwd") else 10) * N * N * d * B * H
                tf = fl * st["n"] / (st["total_ms"] * 1e-3) / 1e12
                roof.update(bound="mfma", achieved=round(tf, 1), peak=MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                            frac=round(tf / MFMA_PEAK_TFLOPS, 4), algo_flops_per_launch=fl,
                            hbm_achieved_gbs=roof["achieved"], hbm_frac=roof["frac"])
        # bytes of one step: measured (counter summaries x launches per step) against algorithmic
        Cq = 3 * C
        rows = B * N
        step_bytes = {
            # q, k, v -> out forward (4 units) + backward (8 units)
            "algo_op_level": 12 * unit_bytes,
            # the module: x (fp32) -> qkv (+ the rounded copy of x the weight gradient reads), core, out -> y; backward:
            # dy -> d out, both weight gradients, d qkv -> dx (fp32)
            "algo_module_level": (rows * (C * 4 + Cq * 2 + C * 2) + 12 * unit_bytes + rows * (C * 2 + C * 2)
                                  + rows * (C * 2 + C * 2) + rows * (C * 2 + C * 2)
                                  + rows * (Cq * 2 + C * 4) + rows * (Cq * 2 + C * 2)),
            "measured_traffic": pmc.get("_step_traffic_bytes"),
            "measured_traffic_attention_kernels": pmc.get("_step_traffic_bytes_attention"),
        }
        cpu = None
        if world == 1 and not a.no_cpu_baseline:
            cpu = cpu_baseline(a.attn, C, H, G)
        # N = 196 and N = 4096 next to the headline N = 784 (north_star): cfg2 at the DeiT batch, cfg5 at the
        # saturating batch 16 and at the faithful batch 1 (--max-tokens 4096)
        others = None
        if world == 1 and a.workload == "cfg3" and a.attn != "causal_eva" and not a.no_other_workloads:
            others = {}
            runs = [("cfg2_N196", a.attn, (128, 192, 3, (14, 14))), ("cfg5_N4096_B16", a.attn, (16, 512, 8, (4096,))),
                    ("cfg5_N4096_B1", a.attn, (1, 512, 8, (4096,)))]
            # the other variants north_star names, at the headline geometry and at N = 4096, so that a driver-run line exists
            # for them too: EVA (named first) and the softmax baseline both are measured against
            for other in ("eva", "softmax"):
                if other != a.attn:
                    runs.append(("cfg3_N784_%s" % other, other, (B, C, H, seq)))
                    if other == "eva":
                        runs.append(("cfg2_N196_eva", "eva", (128, 192, 3, (14, 14))))      # BASELINE.json configs[1]
                    runs.append(("cfg5_N4096_B16_%s" % other, other, (16, 512, 8, (4096,))))
            # cfg4 (BASELINE.json config 4): the attention layers of the four PvTv2-b2 stages at 384 x 384, batch 32 per GPU
            # (pvt_legacy.py:309-319,349-359: dims 64/128/320/512, heads 1/2/5/8; grids 96/48/24/12).  The reference's own
            # recipe (w = 7, 49 landmarks) asserts on these grids: 8 x 8 windows and 36 landmarks are the
            # nearest geometry it runs; the last stage is softmax attention.
            pvt = dict(window_size=8, num_landmarks=36)
            runs += [("cfg4_stage1_N9216_eva", "eva", (32, 64, 1, (96, 96)), pvt), ("cfg4_stage2_N2304_eva", "eva", (32, 128, 2, (48, 48)), pvt),
                     ("cfg4_stage3_N576_eva", "eva", (32, 320, 5, (24, 24)), pvt), ("cfg4_stage4_N144_softmax", "softmax", (32, 512, 8, (12, 12)))]
            # the wikitext-103 decoder self-attention (`--attn causal_eva --workload lm`): the recipe's 18 samples x 512 tokens
            runs.append(("lm_N512_B18_causal_eva", "causal_eva", (18, 1024, 8, (512,))))
            for key, attn_o, (Bo, Co, Ho, so), *ov in runs:
                try:
                    others[key] = measure_workload(attn_o, Bo, Co, Ho, so, dev, tune=tune, overrides=ov[0] if ov else None)
                    others[key]["attn"] = attn_o
                except Exception as ex:          # never let an extra line take the headline measurement down
                    others[key] = {"error": str(ex).split("\n")[0][:200]}
        line = {
            "metric": "attn fwd+bwd tokens/s per GPU at N=784, d=64; 1/2/4/8-GPU DDP scaling",
            "value": value, "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": el / a.steps * 1e3, "eager_ms_per_step": round(eager_ms, 4),
            "eager_ms_per_step_blocks": {"n": len(eager_blocks), "steps_each": n_eager, "min": round(min(eager_blocks), 4),
                                         "median": round(eager_ms, 4), "max": round(max(eager_blocks), 4)},
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "%s attention layer fwd+bwd+SGD, x=[%s] per GPU (N=%d, h=%d, d=%d), "
                                   "bf16 autocast%s" % (a.attn, ",".join(str(v) for v in (B,) + tuple(seq) + (C,)), N, H, d,
                                                        ", data-parallel flat-bucket all-reduce" if ddp else ""),
                       "attn": a.attn, "global_batch": B * world, "seq_len": N, "heads": H, "head_dim": d,
                       "parallelism": "dp%d" % world, "hipgraph": graphed, "ddp_schedule": ddp_scheme,
                       # one flat fp32 bucket of every parameter gradient per step (efficient_attention.data_parallel); a
                       # ring all-reduce moves 2 (N - 1) / N of it over each GPU's links
                       "allreduce_bytes_per_step": (sum(prm.numel() for prm in params) * 4) if ddp else None,
                       "gemm_tunableop": gemm_tuned},
            "hbm_roofline_tokens_per_s_per_gpu": HBM_PEAK_GBS * 1e9 / (BYTES_PER_TOKEN_HEAD * H * d / 64),
            # whole layer priced on the op-level q,k,v -> out traffic (1536*h bytes per token at d = 64)
            "layer_algorithmic_gbs": value / world * BYTES_PER_TOKEN_HEAD * H * d / 64 / 1e9,
            "hbm_measured_copy_gbs": None if copy_gbs is None else round(copy_gbs, 1),
            "hbm_measured_torch_copy_gbs": None if copy_torch_gbs is None else round(copy_torch_gbs, 1),
            "step_bytes": step_bytes,
            "roofline": roof, "cpu_baseline": cpu,
            "ms_per_step_blocks": None if not blocks_ms else {"n": len(blocks_ms), "steps_each": a.steps, "min": round(min(blocks_ms), 4),
                                                              "median": round(sorted(blocks_ms)[len(blocks_ms) // 2], 4),
                                                              "max": round(max(blocks_ms), 4)},
            "other_workloads": others,
            "prewarm_ms": prewarm_ms,
        }
    if ddp:
        dist.destroy_process_group()
    if rank == 0:
        _emit(line)


def main_model(a):
    """Whole-model workloads: one synthetic training step of ea_harness's stacks --
    autocast forward + loss, backward, SGD(momentum) -- timed eagerly (what the reference's call sites
    do) and captured in a hipGraph.  N > 1: DistributedDataParallel over RCCL (bucketed all-reduce
    overlapped with backward), weak scaling."""
    from ea_harness import trainer
    from efficient_attention import _ops
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # EA_BENCH_FORCE_DDP=1 (dev / tests): the N > 1 leg -- RCCL process group, DistributedDataParallel with 25 MB buckets,
    # all-reduce overlapped with backward -- on a single rank, so that it executes on a 1-GPU box
    ddp = world > 1 or bool(os.environ.get("EA_BENCH_FORCE_DDP"))
    if ddp:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29534")
            dist.init_process_group(os.environ.get("EA_BENCH_BACKEND", "nccl"), rank=0, world_size=1)
        else:
            dist.init_process_group(os.environ.get("EA_BENCH_BACKEND", "nccl"))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        wl = trainer.build_workload(a.workload, dev, batch=None if a.batch == 128 else a.batch, seed=1234 + rank)
    if not a.no_gemm_tune:
        import torch.cuda.tunable as tunable
        tunable.enable(True)
        tunable.set_filename(a.gemm_tune_file or os.path.join(tempfile.gettempdir(), "ea_bench_tunableop_%d.csv" % os.getpid()))
        tunable.tuning_enable(not (a.gemm_tune_file and os.path.exists(a.gemm_tune_file)))
    ddp_model = trainer.wrap_ddp(wl.model, dev) if ddp else None
    step = trainer.make_step(wl, ddp_model=ddp_model)
    # what one step puts on the links: every parameter gradient once, fp32, in buckets of <= 25 MB issued as the backward
    # reaches them (ring all-reduce moves 2 (N - 1) / N of that per GPU)
    grad_bytes = sum(p.numel() * 4 for p in wl.model.parameters() if p.requires_grad)
    ddp_info = None
    if ddp:
        ddp_info = {"schedule": "DistributedDataParallel: bucketed all-reduce overlapped with backward, gradients as bucket views",
                    "bucket_cap_mb": 25, "allreduce_bytes_per_step": grad_bytes,
                    "buckets_per_step": max(1, -(-grad_bytes // (25 << 20))),
                    "ring_bytes_on_links_per_gpu": int(2 * (world - 1) / world * grad_bytes), "backend": dist.get_backend()}
    for _ in range(max(a.warmup, 2)):
        step()
    torch.cuda.synchronize()
    if not a.no_gemm_tune:
        tunable.tuning_enable(False)

    def timed(fn, n):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    for _ in range(a.warmup):
        step()
    eager_el = timed(step, a.steps)
    graph_ms = None
    if not a.no_graph and not ddp:
        try:
            replay = trainer.capture_step(step)
            for _ in range(3):
                replay()
            graph_ms = timed(replay, a.steps) / a.steps * 1e3
        except Exception as ex:
            print("graph capture unavailable (%s)" % str(ex).split("\n")[0], file=sys.stderr)
            torch.cuda.synchronize()
    _ops.KERNEL_TIMER.enable()
    for _ in range(min(a.steps, 5)):
        step()
    torch.cuda.synchronize()
    ktimes = _ops.KERNEL_TIMER.summary()
    _ops.KERNEL_TIMER.disable()
    if rank == 0:
        tokens = wl.tokens * world
        hip_ms = sum(v["total_ms"] for v in ktimes.values()) / max(min(a.steps, 5), 1)
        line = {
            "metric": "attn fwd+bwd tokens/s per GPU at N=784, d=64; 1/2/4/8-GPU DDP scaling",
            "value": tokens * a.steps / eager_el, "unit": "tokens/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": eager_el / a.steps * 1e3, "eager_ms_per_step": eager_el / a.steps * 1e3,
            "graph_ms_per_step": graph_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "%s: %s; whole-model training step (fwd + loss + bwd + SGD momentum), eager, bf16 "
                                   "autocast%s" % (a.workload, wl.desc, ", DistributedDataParallel" if ddp else ""),
                       "tokens_per_step_per_gpu": wl.tokens, "parallelism": "dp%d" % world, "hipgraph": False,
                       "ddp_schedule": ddp_info,
                       "params_M": round(sum(p.numel() for p in wl.model.parameters()) / 1e6, 2)},
            "attention_core_ms_per_step": round(hip_ms, 4),
            "attention_kernels_avg_us": {k: round(v["avg_ms"] * 1e3, 2) for k, v in ktimes.items()},
            "roofline": None, "cpu_baseline": None,
        }
    if ddp:
        dist.destroy_process_group()
    if rank == 0:
        _emit(line)


if __name__ == "__main__":
    main()
