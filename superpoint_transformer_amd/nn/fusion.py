"""Feature fusion modules (src/nn/fusion.py): cat / additive / first / second."""
import torch
from torch import nn

__all__ = ["CatFusion", "AdditiveFusion", "TakeFirstFusion", "TakeSecondFusion",
           "fusion_factory"]


class _Fusion(nn.Module):
    def forward(self, x1, x2):
        if x1 is None:
            return x2
        if x2 is None:
            return x1
        return self._fuse(x1, x2)


class CatFusion(_Fusion):
    def _fuse(self, x1, x2):
        return torch.cat((x1, x2), dim=1)


class AdditiveFusion(_Fusion):
    def _fuse(self, x1, x2):
        return x1 + x2


class TakeFirstFusion(_Fusion):
    def _fuse(self, x1, x2):
        return x1


class TakeSecondFusion(_Fusion):
    def _fuse(self, x1, x2):
        return x2


def fusion_factory(mode):
    if mode in ("cat", "concatenate", "concatenation", "|"):
        return CatFusion()
    if mode in ("residual", "additive", "+"):
        return AdditiveFusion()
    if mode in ("first", "1", "1st"):
        return TakeFirstFusion()
    if mode in ("second", "2", "2nd"):
        return TakeSecondFusion()
    raise NotImplementedError(f"Unknown mode='{mode}'")
