"""Factory entries of the reference that are outside this build's hot-path scope (SURVEY.md 8:
`ra` and `scatterbrain` are marked "next").  They keep the names importable and fail
loudly on construction instead of silently running something else."""
import torch.nn as nn


class _Unported(nn.Module):
    _what = ""

    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError(
            "%s is not part of the MI355X hot-path build yet (SURVEY.md 8f)" % self._what)


class RandomizedAttention(_Unported):
    _what = "RandomizedAttention ('ra')"


class ScatterBrain(_Unported):
    _what = "ScatterBrain ('scatterbrain')"
