"""Drop-in for reference ``mhmocap/morphology.py`` (``Erode2D`` / ``Dilate2D`` / ``BinaryMorphology`` on
(B,C,H,W) float maps, threshold 0.5, square window, zero padding) backed by ``mh_morph_f32``.  Inside the
optimiser the erosion of the constant instance masks runs once at staging time on the packed bit planes
(``mh_erode_bits``); these modules exist for callers that use them directly."""
import torch
from torch import nn

from mhhip import _lib
from mhhip._lib import check, ptr

_MODES = {'erode': 0, 'dilate': 1}


def _morph(x, window, mode):
    """binary erosion / dilation of every (H, W) plane of x on the device; returns a new float tensor"""
    planes = x.contiguous().float()
    result = torch.empty_like(planes)
    rows, cols = planes.shape[-2], planes.shape[-1]
    check(_lib.lib().mh_morph_f32(ptr(planes), ptr(result), planes.numel() // (rows * cols), rows, cols, int(window),
                                  _MODES[mode], _lib.stream_ptr(planes.device)))
    return result


class BinaryMorphology(nn.Module):
    """reference morphology.py:6-33 (same constructor arguments and assertion)"""

    def __init__(self, kernel_size=5, type=None):
        nn.Module.__init__(self)
        assert type in ['dilate', 'erode'], (f'Invalid `type` {type}')
        self.opp_type, self.kernel_size = type, kernel_size

    def forward(self, x):
        return _morph(x, self.kernel_size, self.opp_type)


class Dilate2D(BinaryMorphology):
    """reference morphology.py:35-37"""

    def __init__(self, kernel_size=5):
        BinaryMorphology.__init__(self, kernel_size=kernel_size, type='dilate')


class Erode2D(BinaryMorphology):
    """reference morphology.py:39-41"""

    def __init__(self, kernel_size=5):
        BinaryMorphology.__init__(self, kernel_size=kernel_size, type='erode')


# names of the shadowed reference module this file does not define (mhhip/_overlay.py)
from mhhip._overlay import inherit as _inherit  # noqa: E402

_inherit(globals())
