import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "efficient-attention_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# dev switches that turn a kernel family OFF (tools/switch_matrix.sh runs the suite once per switch): the tests OF that family --
# its own parity tests and the fixtures only it can run -- are skipped, everything else must stay green on the other path
SWITCHED_OFF = {
    "EA_DGRAD_RS": (lambda it: "linear_dgrad" in it.name, "ea_linear_dgrad* are switched off (EA_DGRAD_RS=0)"),
    "EA_F32_CORES": (lambda it: it.fspath.basename == "test_gpu_f32_cores.py" or "eva_2d_L100" in it.name
                     or "lara_2d_L144" in it.name or "many_chunks" in it.name,
                     "the fp32 cores are switched off (EA_F32_CORES=0): fp32-path tests and the geometries only they cover"),
    # the single-node module paths are what these two tests compare against the three-node / dense-bias paths
    "EA_EVA_MODULE_FN": (lambda it: ("test_wide_module_path" in it.name or "test_table_bias_module_path" in it.name) and "eva" in it.name,
                         "EvaModuleFn is switched off (EA_EVA_MODULE_FN=0)"),
    # (USE_LARA_MODULE_FN gates lara_module_fn_supported, which every single-node path asks)
    "EA_LARA_MODULE_FN": (lambda it: "test_wide_module_path" in it.name, "the single-node paths are switched off (EA_LARA_MODULE_FN=0)"),
    "EA_CORE_MODULE_FN": (lambda it: ("test_wide_module_path" in it.name and ("softmax" in it.name or "local" in it.name))
                          or ("test_table_bias_module_path" in it.name and "local" in it.name),
                          "CoreModuleFn is switched off (EA_CORE_MODULE_FN=0)"),
}


def pytest_collection_modifyitems(config, items):
    import torch
    for env, (match, why) in SWITCHED_OFF.items():
        if os.environ.get(env, "1") == "0":
            mark = pytest.mark.skip(reason=why)
            for item in items:
                if match(item):
                    item.add_marker(mark)
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
