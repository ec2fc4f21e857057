"""efficient_attention -- MI355X-native drop-in for HKUNLP/efficient-attention's attention library.

Same import name and plugin surface as the reference package
(efficient-attention/efficient_attention/__init__.py:5-79): `AttentionFactory`,
`NestedNamespace`, `add_nested_argument`, `remove_argument`, `remove_prefix`, and nn.Modules
with the reference's constructor kwargs, `forward(x, key_padding_mask=None)` protocol and
state_dict keys -- so `vit/models/efficient_vit.py:112`, `vit/models/pvt_legacy.py:85` and
`fairseq/fairseq/modules/efficient_attention.py:64` work unchanged.  What differs is below the
modules: every q,k,v -> out core is a hand-written HIP kernel for gfx950 in libea_hip.so
(include/ea_hip.h), reached through ctypes (`_native.py`).  There is no CPU fallback.
"""
import argparse
from typing import Dict


def remove_argument(parser, arg):
    """Drop option `arg` (flag string or dest) from an argparse parser (reference :6-17)."""
    for action in list(parser._actions):
        flags = action.option_strings
        if (flags and flags[0] == arg) or action.dest == arg:
            parser._remove_action(action)
            break
    for group in parser._action_groups:
        for action in list(group._group_actions):
            if action.dest == arg:
                group._group_actions.remove(action)
                return


def remove_prefix(text, prefix):
    return text[len(prefix):] if text.startswith(prefix) else text


def add_nested_argument(parser, name, struct_name="attn_args", prefix="", **kwargs):
    """add_argument whose dest is '<struct_name>.<flag>' so NestedNamespace groups it
    (reference :22-27).  With prefix='encoder-attn', '--encoder-attn-window-size' lands in
    '<struct_name>.window_size'."""
    flag = name.lstrip("-") if not prefix else remove_prefix(name, "--%s-" % prefix)
    parser.add_argument(name, dest="%s.%s" % (struct_name, flag.replace("-", "_")), **kwargs)


class NestedNamespace(argparse.Namespace):
    """Namespace in which setting 'a.b' creates/extends a nested namespace `a` (reference :31-39)."""

    def __setattr__(self, name, value):
        if "." not in name:
            self.__dict__[name] = value
            return
        head, rest = name.split(".", 1)
        child = getattr(self, head, None)
        if child is None:
            child = NestedNamespace()
        setattr(child, rest, value)
        self.__dict__[head] = child


from . import _dispatch  # noqa: E402,F401  (registers torch.ops.ea.*)
from .abstract_attention import MultiheadAttention  # noqa: E402
from .local_attention import LocalAttention  # noqa: E402
from .kernelized_attention import KernelizedAttention  # noqa: E402
from .lara import LinearRA  # noqa: E402
from .eva import EVA  # noqa: E402
from .causal_eva import CausalEVAttention  # noqa: E402
from .randomized_attention import RandomizedAttention  # noqa: E402
from .scatterbrain_attention import ScatterBrain  # noqa: E402


class AttentionFactory(object):
    """name -> module class registry (reference :52-79)."""

    attn_dict = {
        "performer": KernelizedAttention,
        "softmax": MultiheadAttention,
        "local": LocalAttention,
        "lara": LinearRA,
        "ra": RandomizedAttention,
        "scatterbrain": ScatterBrain,
        "eva": EVA,
        "causal_eva": CausalEVAttention,
    }

    @classmethod
    def build_attention(cls, attn_name: str, attn_args: Dict):
        return cls.attn_dict[attn_name](**attn_args)       # KeyError / TypeError as the reference

    @classmethod
    def add_attn_specific_args(cls, parent_parser, attn_name, struct_name="attn_args", prefix=""):
        attn_cls = cls.attn_dict[attn_name]
        if hasattr(attn_cls, "add_attn_specific_args"):
            return attn_cls.add_attn_specific_args(parent_parser, struct_name=struct_name, prefix=prefix)
        return parent_parser


__all__ = ["AttentionFactory", "NestedNamespace", "add_nested_argument", "remove_argument",
           "remove_prefix", "MultiheadAttention", "LocalAttention", "KernelizedAttention",
           "LinearRA", "EVA", "RandomizedAttention", "ScatterBrain", "CausalEVAttention"]
