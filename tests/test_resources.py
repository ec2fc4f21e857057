"""-m "not gpu": register / scratch budget of the shipped kernels, read from the code-object metadata of libea_hip.so
(tools/resource_report.py).  No kernel of the library may use scratch memory: a change that
makes one spill fails here instead of showing up as a slowdown."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

LLVM = os.environ.get("EA_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-objdump")) or shutil.which("c++filt") is None,
                                reason="needs llvm-objdump / llvm-readelf / c++filt")

# Round 6: NO kernel of the library spills (36 did at the end of round 5: the run-time-geometry window kernels, the finish pass
# of the qkv input gradient, two LARA backward instantiations -- every one of them lane-derived loop invariants kept alive
# across a loop or a phase boundary; they are re-derived from an opaque copy of the thread index where they are used).
KNOWN_SPILLS = {}
# the launches of the default bench step (LARA, cfg3, bf16) that must stay spill-free
HEADLINE = ["proj_rs_kernel<BF16, true, 16, true>", "proj_rs_kernel<BF16, true, 16, false>", "lmk2::lmk2_kernel<64, false>", "lmk2::lmk2_kernel<64, true>", "lara_y_kernel<BF16, 64, 0, 0>",
            "lara_x_kernel<BF16, 64, 4, 7, 0>", "lara_fq_kernel<BF16, 64, 4, 1, 0>", "wgrad_kernel<BF16, 192, 192>", "dgrad_rs_kernel<BF16, false, true>",
            "lin_kernel<BF16, 6, 2, false, 6, false>"]


def test_scratch_budget_of_the_library():
    import resource_report as rr
    rows = rr.kernels()
    assert len(rows) > 500
    short = {rr._short(r["demangled"]): r for r in rows}
    for name in HEADLINE:
        assert name in short, name
        assert short[name]["scratch"] == 0, (name, short[name])
    for name, r in short.items():
        if not r["scratch"]:
            continue
        fam = [f for f in KNOWN_SPILLS if name.startswith(f)]
        assert fam, "a kernel outside the known families spills: %s (%d B/lane)" % (name, r["scratch"])
        assert r["scratch"] <= KNOWN_SPILLS[fam[0]], (name, r["scratch"])
