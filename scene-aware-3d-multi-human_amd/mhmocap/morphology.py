"""Drop-in for reference ``mhmocap/morphology.py`` (``Erode2D`` / ``Dilate2D`` on (B,C,H,W) float
maps) backed by ``mh_morph_f32``.  Inside the optimiser the erosion of the constant instance masks
runs once at staging time on the packed bit planes (``mh_erode_bits``)."""
import torch
import torch.nn as nn

from mhhip import _lib
from mhhip._lib import check, ptr


class BinaryMorphology(nn.Module):
    def __init__(self, kernel_size=5, type=None):
        super().__init__()
        assert type in ['dilate', 'erode'], (f'Invalid `type` {type}')
        self.kernel_size = kernel_size
        self.opp_type = type

    def forward(self, x):
        x = x.contiguous().float()
        out = torch.empty_like(x)
        H, W = x.shape[-2:]
        check(_lib.lib().mh_morph_f32(ptr(x), ptr(out), x.numel() // (H * W), H, W, int(self.kernel_size),
                                      1 if self.opp_type == 'dilate' else 0, _lib.stream_ptr(x.device)))
        return out


class Dilate2D(BinaryMorphology):
    def __init__(self, kernel_size=5):
        super().__init__(kernel_size, 'dilate')


class Erode2D(BinaryMorphology):
    def __init__(self, kernel_size=5):
        super().__init__(kernel_size, 'erode')
