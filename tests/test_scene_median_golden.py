"""The per-pixel masked median over time of the scene update (reference fhsog.py:180-202, called from
optimizer.py:579-582) against the reference's own output (tests/golden: ``median_*`` on the T=20 fit inputs, where
every pixel is seen; ``median2_*`` with ties, never-seen pixels and pixels seen by 1 / 2 / 4 frames): the numpy
checker (oracle/scene_oracle.py) on the CPU, and the two device kernels
(``mh_scene_median`` / ``mh_scene_median_t``) on the GPU."""
import numpy as np
import pytest
import torch

import golden_inputs as gi


def _case(which, golden, golden_raster):
    if which == 1:
        fin = gi.fit_inputs()
        return fin['depths'], fin['backmasks'], fin['images'], golden['median_img'], golden['median_depth'], golden['median_mask']
    dn, back, imgs = gi.median_inputs()
    return dn, back, imgs, golden_raster['median2_img'], golden_raster['median2_depth'], golden_raster['median2_mask']


@pytest.mark.parametrize('which', [1, 2])
def test_host_median_matches_reference(golden, golden_raster, which):
    from oracle import scene_oracle as scene_host
    dn, back, imgs, wimg, wdep, wmask = _case(which, golden, golden_raster)
    img, dep, msk = scene_host.aggregate_scene_median((1.0 / (dn + 0.5)).astype(np.float32), imgs, back)
    np.testing.assert_array_equal(msk, wmask)
    np.testing.assert_array_equal(img[msk], wimg[msk])
    np.testing.assert_allclose(dep[msk], wdep[msk], rtol=1e-6)
    if which == 2:
        assert (~wmask).sum() >= 7 and wmask.sum() > 200          # never-seen pixels are part of the fixture


@pytest.mark.gpu
@pytest.mark.parametrize('which', [1, 2])
def test_device_median_matches_reference(golden, golden_raster, which):
    from mhhip import _lib as _l
    L = _l.lib()
    dn, back, imgs, wimg, want, wmask = _case(which, golden, golden_raster)
    T, H, W = dn.shape
    dev = torch.device('cuda:0')
    # the kernel rebuilds 1/target_disp from the normalised disparity and the depth-range leaves (optimizer.py:425-426):
    # depth = 1/(d*(1/min_z - 1/max_z) + 1/max_z) with min_z = 2/3, max_z = 2  ->  1/(d + 0.5), the fixture's input
    sp_inv = lambda y: float(np.log(np.expm1(y)))
    zmin = np.full(T, sp_inv(2.0 / 3.0), np.float32)
    zmax = np.full(T, sp_inv(2.0 - 2.0 / 3.0 - 1.0), np.float32)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), device=dev)
    tdn, tback, tzmin, tzmax = t(dn), t(back.astype(np.uint8)), t(zmin), t(zmax)
    ws = torch.empty(L.mh_scene_workspace_bytes(T, H, W), dtype=torch.uint8, device=dev)
    md, mm = torch.empty(H, W, device=dev), torch.empty(H, W, device=dev)
    _l.check(L.mh_scene_median(T, H, W, _l.ptr(tdn), _l.ptr(tback), _l.ptr(tzmin), _l.ptr(tzmax), _l.ptr(md), _l.ptr(mm),
                               _l.ptr(ws), _l.stream_ptr(dev)))
    md2, mm2 = torch.empty(H, W, device=dev), torch.empty(H, W, device=dev)
    tdn_t, tback_t = tdn.view(T, H * W).t().contiguous(), tback.view(T, H * W).t().contiguous()
    _l.check(L.mh_scene_median_t(T, H, W, _l.ptr(tdn_t), _l.ptr(tback_t), _l.ptr(tzmin), _l.ptr(tzmax), _l.ptr(md2),
                                 _l.ptr(mm2), _l.ptr(ws), _l.stream_ptr(dev)))
    torch.cuda.synchronize()
    for d, m in ((md, mm), (md2, mm2)):
        wm = np.asarray(wmask, bool)
        np.testing.assert_array_equal(m.cpu().numpy() > 0.5, wm)
        # the leaves reproduce (1/min_z - 1/max_z, 1/max_z) = (1, 0.5) to a few ulp: 3e-6 relative on the depth
        np.testing.assert_allclose(d.cpu().numpy()[wm], want[wm], rtol=3e-6)
