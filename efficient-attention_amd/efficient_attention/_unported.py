"""Factory entries of the reference that are outside this build's hot-path scope (SURVEY.md 8:
`scatterbrain` is marked "next").  It keeps the name importable and fails
loudly on construction instead of silently running something else."""
import torch.nn as nn


class _Unported(nn.Module):
    _what = ""

    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError(
            "%s is not part of the MI355X hot-path build yet (SURVEY.md 8f)" % self._what)


class ScatterBrain(_Unported):
    _what = "ScatterBrain ('scatterbrain')"
