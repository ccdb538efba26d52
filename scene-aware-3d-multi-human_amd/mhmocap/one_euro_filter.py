"""Host-side one-euro filter with the interface of reference ``mhmocap/one_euro_filter.py:17-53``
(used by ``get_filtered_vertices_by_smpl`` and the data pre-processing).  The per-cycle filtering of
the optimiser runs on the GPU (``mh_one_euro_scan``)."""
import math

import numpy as np


def _alpha(t_e, cutoff):
    r = 2 * math.pi * cutoff * t_e
    return r / (r + 1)


class OneEuroFilter(object):
    def __init__(self, t0, x0, dx0=None, min_cutoff=0.004, beta=0.7, d_cutoff=1.0):
        self.min_cutoff, self.beta, self.d_cutoff = float(min_cutoff), float(beta), float(d_cutoff)
        if dx0 is None:
            dx0 = np.zeros_like(x0)
        self.x_prev, self.dx_prev, self.t_prev = x0, dx0, t0

    def __call__(self, t, x, mask=None):
        if mask is None:
            mask = np.ones_like(x)
        t_e = t - self.t_prev
        a_d = _alpha(t_e, self.d_cutoff)
        dx_hat = a_d * ((x - self.x_prev) / t_e) + (1 - a_d) * self.dx_prev
        a = _alpha(t_e, self.min_cutoff + self.beta * np.abs(dx_hat))
        x_hat = a * x + (1 - a) * self.x_prev
        keep = 1 - mask
        self.x_prev = keep * self.x_prev + mask * x_hat
        self.dx_prev = keep * self.dx_prev + mask * dx_hat
        self.t_prev = keep * self.t_prev + mask * t
        return keep * x + mask * x_hat


# names of the shadowed reference module this file does not define (mhhip/_overlay.py)
from mhhip._overlay import inherit as _inherit  # noqa: E402

_inherit(globals())
