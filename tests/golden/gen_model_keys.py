#!/usr/bin/env python3
"""Dump the state_dict key -> shape tables of the REFERENCE's whole models into
tests/golden/model_keys.json (data only), for the checkpoint-compatibility test of ea_harness.

Runs only in the build container: imports /root/reference/vit/models/{efficient_vit,pvt_legacy}.py with
throw-away stand-ins for the parts of `timm` those two files import (DropPath, to_2tuple,
trunc_normal_, register_model, _cfg) -- none of which owns a parameter."""
import argparse
import importlib
import json
import os
import sys
import types

import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def install_shims():
    timm = types.ModuleType("timm")
    models = types.ModuleType("timm.models")
    layers = types.ModuleType("timm.models.layers")
    registry = types.ModuleType("timm.models.registry")
    vt = types.ModuleType("timm.models.vision_transformer")

    class DropPath(nn.Module):
        def __init__(self, p=0.):
            super().__init__()
            self.p = p

        def forward(self, x):
            return x
    layers.DropPath = DropPath
    layers.trunc_normal_ = torch.nn.init.trunc_normal_
    layers.to_2tuple = lambda v: v if isinstance(v, tuple) else (v, v)
    registry.register_model = lambda f: f
    vt._cfg = lambda **kw: {}
    timm.models = models
    models.layers, models.registry, models.vision_transformer = layers, registry, vt
    for name, mod in (("timm", timm), ("timm.models", models), ("timm.models.layers", layers),
                      ("timm.models.registry", registry), ("timm.models.vision_transformer", vt)):
        sys.modules[name] = mod
    sys.path.insert(0, os.path.join(REF, "efficient-attention"))
    pkg = types.ModuleType("refvit")
    pkg.__path__ = [os.path.join(REF, "vit")]
    sys.modules["refvit"] = pkg
    mpkg = types.ModuleType("refvit.models")
    mpkg.__path__ = [os.path.join(REF, "vit", "models")]
    sys.modules["refvit.models"] = mpkg


def table(model):
    return {k: list(v.shape) for k, v in model.state_dict().items()}


def main():
    install_shims()
    ev = importlib.import_module("refvit.models.efficient_vit")
    pv = importlib.import_module("refvit.models.pvt_legacy")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "efficient-attention_amd"))
    out = {}

    def ns(**kw):
        return argparse.Namespace(**kw)
    eva = dict(fp32=False, use_rpe=True, window_size=7, attn_2d=True, overlap_window=False, adaptive_proj="default",
               num_landmarks=49, use_t5_rpe=False)
    lara = dict(fp32=False, num_landmarks=49, kernel_size=None, pool_module_type="light", mis_type="mis-opt",
                proposal_gen="pool-mixed", use_antithetics=False, use_multisample=False, alpha_coeff=2.0)
    for name, patch, attn, aargs in (("deit_tiny_p16_eva", 16, "eva", eva), ("deit_tiny_p8_lara", 8, "lara", lara)):
        a = ns(num_classes=1000, input_size=224, patchify_stem="default", no_pos_emb=False, drop_rate=0.0,
               attn_drop_rate=0.0, drop_path_rate=0.1, use_glu=False, num_heads=None, attn_name=attn,
               attn_specific_args=ns(**aargs))
        fn = ev.evit_tiny_p16 if patch == 16 else ev.evit_tiny_p8
        out[name] = table(fn(a))
    eva8 = dict(eva, window_size=8, num_landmarks=36)
    a = ns(num_classes=1000, input_size=384, drop_rate=0.0, attn_drop_rate=0.0, use_conv_patchify=False,
           attn_name="eva", attn_specific_args=ns(**eva8))
    out["pvt_b2_eva"] = table(pv.pvt_small(a))
    with open(os.path.join(HERE, "model_keys.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print({k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
