"""The HIP rasteriser against PyTorch3D's own outputs -- ACTIVE ONLY when tests/golden/reference_pytorch3d.npz exists (see
tests/test_oracle_golden_pytorch3d.py and tests/golden/make_golden_pytorch3d.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_pytorch3d.npz')


@pytest.mark.skipif(not os.path.exists(FIX), reason='tests/golden/reference_pytorch3d.npz not generated yet')
@pytest.mark.parametrize('name', ['cross', 'por', 'sq'])
def test_hip_depth_and_silhouette_match_pytorch3d(smpl_struct, smpl_regs, name):
    from mhhip import engine
    from mhhip.raster import render
    p3d = np.load(FIX)
    g = lambda k: p3d[name + '_' + k]
    W, H = [int(x) for x in g('size')]
    model = engine.BodyModel(smpl_struct, smpl_regs)
    zb, al = render(model, torch.tensor(g('verts')).cuda(), g('cam_K'), (W, H))
    zb, al = zb.cpu().numpy(), al.cpu().numpy()
    want = g('zbuf8')
    both = (zb > 0) & (want > 0)
    assert ((zb > 0) == (want > 0)).mean() > 0.9995
    # the nearest face of a pixel may be a float32 near-tie; where the depth agrees it must agree tightly
    close = np.abs(zb - want)[both] <= 2e-5 * want[both]
    assert close.mean() > 0.995
    assert (np.abs(al - g('alpha')) <= 5e-4).mean() > 0.995
