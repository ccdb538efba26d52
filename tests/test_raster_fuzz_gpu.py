"""The selection kernel against the oracle's brute-force selection on random scenes: image shapes (landscape, portrait,
square, the bench's 240x135), 1-3 bodies, close-ups (faces of many pixels: the kernel's general path) to distant bodies
(sub-pixel faces: its culled path), three fields of view.  Every pixel whose face lists differ must be a float64 near-tie
in the sense of tests/test_raster_gpu.py::_selection_differences.  (tools/fuzz_raster.py runs the same loop for any number
of scenes: 300 of them had no real difference.)"""
import numpy as np
import pytest

import test_raster_gpu as tr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('seed', [101, 202])
def test_random_scenes_differ_from_the_brute_force_selection_only_in_near_ties(smpl_struct, smpl_regs, seed):
    rng = np.random.RandomState(seed)
    total_live = 0
    for c in range(8):
        W, H = [(96, 54), (64, 96), (80, 80), (160, 90), (48, 135), (240, 135)][rng.randint(6)]
        T, N = int(rng.randint(1, 3)), int(rng.randint(1, 4))
        zlo = float(rng.choice([1.1, 1.6, 2.5, 4.0]))
        zhi = zlo + float(rng.choice([0.3, 1.0, 3.0]))
        fov = float(rng.choice([40.0, 60.0, 90.0]))
        r = tr._run_case(smpl_struct, smpl_regs, T, N, W, H, int(rng.randint(1 << 30)), zlo=zlo, zhi=zhi, fov=fov)
        ndiff, live, not_ties = tr._selection_differences(r)
        total_live += live
        assert not not_ties, 'scene %d (%dx%d, T %d, N %d, z %.1f-%.1f, fov %.0f): %s' % (c, W, H, T, N, zlo, zhi, fov, not_ties[:4])
        assert ndiff <= 0.04 * max(live, 1) + 4, (ndiff, live)
    assert total_live > 5000
