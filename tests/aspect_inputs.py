"""the portrait / square raster fixture (tests/golden/make_golden_raster_aspect.py) as per-variant dicts with the key layout of
reference_raster_cpu.npz"""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = ['por', 'sq']


def load(tag):
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'reference_raster_aspect_cpu.npz'), allow_pickle=False)
    return {k[len(tag) + 1:]: z[k] for k in z.files if k.startswith(tag + '_')}
