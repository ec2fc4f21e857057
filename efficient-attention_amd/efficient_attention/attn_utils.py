"""Small host-side helpers kept for import compatibility with the reference's
efficient_attention/attn_utils.py.  The windowing functions of the reference
(window_{1d,2d}_partition/merge, pad_to_multiple on q/k/v) have no counterpart here: the HIP
kernels do that index arithmetic in place."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class FlattenTranspose(nn.Module):
    """[b, c, H, W] -> [b, H*W, c] (reference attn_utils.py:92-96); LARA's pooling stack uses it."""

    def forward(self, x):
        return x.flatten(2).transpose(1, 2)


def pad_to_multiple(tensor, multiple, dim=-2, value=0, create_mask=False):
    """Right-pad `dim` (negative index) to a multiple; optionally return the [B, n] pad mask
    (reference attn_utils.py:12-30)."""
    assert dim < 0
    n = int(tensor.shape[dim])
    extra = (-n) % multiple
    if extra:
        tensor = F.pad(tensor, (0, 0) * (-1 - dim) + (0, extra), value=value)
    if not create_mask:
        return tensor
    mask = torch.zeros(tensor.shape[0], tensor.shape[-2], dtype=torch.bool, device=tensor.device)
    if extra:
        mask[:, -extra:] = True
    return tensor, mask
