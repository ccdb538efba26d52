"""Rows a13 / a14 against PyTorch3D itself -- ACTIVE ONLY when tests/golden/reference_pytorch3d.npz exists.

PyTorch3D is installed neither in the build container nor on the GPU image, so the rasteriser inside the two rasterised terms
is "parity unpinned" (DESIGN.md section 6).  tests/golden/make_golden_pytorch3d.py produces the fixture on any machine that
has PyTorch3D (one command, no reference, no GPU); with the file in place these tests hold the oracle's restatement
(oracle/raster_oracle.py + oracle/raster_select.c) against PyTorch3D's own fragments, silhouette and gradient, and
tests/test_pytorch3d_golden_gpu.py does the same for the HIP kernels.  Without the file: one test checks that the generator's
scenes can still be built from the committed fixtures, the rest skip."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from oracle import raster_oracle as ro

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = os.path.join(HERE, 'golden', 'reference_pytorch3d.npz')
needs_fixture = pytest.mark.skipif(not os.path.exists(FIX), reason='tests/golden/reference_pytorch3d.npz not generated yet '
                                   '(python tests/golden/make_golden_pytorch3d.py where PyTorch3D is installed)')


def _generator():
    spec = importlib.util.spec_from_file_location('make_golden_pytorch3d', os.path.join(HERE, 'golden', 'make_golden_pytorch3d.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_generator_scenes_build_from_the_committed_fixtures():
    m = _generator()
    sc = m.scenes()
    assert [s[0] for s in sc] == ['cross', 'por', 'sq']
    for name, verts, faces, K, (W, H) in sc:
        assert verts.ndim == 3 and verts.shape[2] == 3 and np.isfinite(verts).all()
        ndc = ro.to_ndc(torch.tensor(verts[:1]), K, (W, H)).numpy()
        assert (np.abs(ndc[..., :2]) < 2.5).mean() > 0.9, name      # the bodies are on screen
    # the calibration the script hands to FoVPerspectiveCameras is the one the oracle projects with
    for size in ((96, 60), (54, 90), (64, 64)):
        K = np.array([[80., 0, size[0] / 2 + 1.5], [0, 82., size[1] / 2 - 0.5], [0, 0, 1]], np.float32)
        P = m.calibration(1.0, 100.0, K, size)
        v = torch.tensor([[[0.3, -0.2, 3.0], [-0.5, 0.4, 5.0]]])
        want = ro.to_ndc(v, K, size).numpy()[0]
        hom = np.concatenate([v.numpy()[0] * np.array([-1, -1, 1], np.float32), np.ones((2, 1), np.float32)], 1) @ P.T
        np.testing.assert_allclose(hom[:, :2] / hom[:, 3:4], want[:, :2], rtol=1e-6, atol=1e-7)


@pytest.fixture(scope='module')
def p3d():
    return dict(np.load(FIX))


def _scene(p3d, name):
    g = lambda k: p3d[name + '_' + k]
    W, H = [int(x) for x in g('size')]
    return g, W, H


@needs_fixture
@pytest.mark.parametrize('name', ['cross', 'por', 'sq'])
def test_selection_and_fragments_match_pytorch3d(p3d, smpl_struct, name):
    g, W, H = _scene(p3d, name)
    faces = np.asarray(smpl_struct.f).astype(np.int64)
    verts = torch.tensor(g('verts'))
    ndc = ro.to_ndc(verts, g('cam_K'), (W, H)).numpy().astype(np.float32)
    f8, _ = ro.select_faces(ndc, faces, H, W, 1e-4, 8)
    f4, _ = ro.select_faces(ndc, faces, H, W, 2e-5, 4)
    live = g('p2f8_0') >= 0
    assert live.sum() > 200
    assert ((f8[..., 0] >= 0) == live).mean() > 0.9995
    assert (f8[..., 0] == g('p2f8_0'))[live].mean() > 0.995          # float32 near-ties may be resolved either way
    same4 = (np.sort(f4, -1) == np.sort(g('p2f4'), -1)).all(-1)
    assert same4[(g('p2f4') >= 0).any(-1)].mean() > 0.99
    z, d, valid = ro.fragments(torch.tensor(ndc), faces, g('p2f4'), H, W)           # the oracle's values ON PyTorch3D's faces
    m = g('p2f4') >= 0
    np.testing.assert_allclose(z.numpy()[m], g('z4')[m], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(d.numpy()[m], g('d4')[m], rtol=1e-3, atol=2e-9)


@needs_fixture
@pytest.mark.parametrize('name', ['cross', 'por', 'sq'])
def test_depth_silhouette_and_gradient_match_pytorch3d(p3d, smpl_struct, name):
    from parity_gates import two_precision_gate
    g, W, H = _scene(p3d, name)
    faces = np.asarray(smpl_struct.f).astype(np.int64)
    sel = (g('p2f8_0')[..., None], g('p2f4'))
    outs = {}
    for dt in (torch.float32, torch.float64):
        v = torch.tensor(g('verts')).to(dt).requires_grad_(True)
        zb, al = ro.render(v, faces, g('cam_K'), (W, H), selection=sel)
        (torch.tensor(g('wz')).to(dt) * torch.where(zb > 0, zb, torch.zeros_like(zb))).sum().add((torch.tensor(g('wa')).to(dt) * al).sum()).backward()
        outs[dt] = (zb.detach().numpy(), al.detach().numpy(), v.grad.numpy())
    zb, al, _ = outs[torch.float32]
    np.testing.assert_allclose(np.where(zb > 0, zb, -1), g('zbuf8'), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(al, g('alpha'), atol=2e-4)
    # PyTorch3D's float32 backward against the oracle's autograd at both precisions
    two_precision_gate(g('g_verts'), outs[torch.float32][2], outs[torch.float64][2], 5e-4, name)
