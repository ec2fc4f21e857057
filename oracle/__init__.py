"""oracle/ -- CPU restatement of the reference's attention hot path.  TEST INFRASTRUCTURE ONLY.

This package is the checker for the HIP path, never the thing shipped or measured:
only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import it.  The
product package (efficient-attention_amd/efficient_attention) never imports it and fails
loudly when the HIP library is missing or a CPU tensor reaches an attention core.

Language: PyTorch on CPU (fp32 or fp64) -- the path is floating point, so autograd of the
restatement is also the reference for every backward kernel.  Each function cites the
reference file:line it restates (paths relative to
/root/reference/efficient-attention/efficient_attention/).

Pinning: the reference's own tests hold NO golden vectors for this path, so
the oracle is pinned against outputs of the reference itself, generated in the build
container by tests/golden/gen_golden.py (which imports the reference) and committed as
tests/golden/*.npz.  tests/test_oracle_golden.py checks y, dL/dx and every parameter
gradient for all cases in eval and training mode (injected noise) at fp32 tolerance.
"""
from .geometry import (window_index_1d, window_index_2d, rpe_index_2d, t5_bucket,
                       adaptive_pool_matrix, causal_window_index_1d, t5_bucket_causal)
from .attention import (softmax_core, local_core, eva_core, lara_core, performer_core,
                        causal_eva_core, ra_core, scatterbrain_core, module_forward, default_args)

__all__ = ["window_index_1d", "window_index_2d", "rpe_index_2d", "t5_bucket",
           "adaptive_pool_matrix", "causal_window_index_1d", "t5_bucket_causal", "softmax_core",
           "local_core", "eva_core", "lara_core", "performer_core", "causal_eva_core", "ra_core", "scatterbrain_core",
           "module_forward", "default_args"]
