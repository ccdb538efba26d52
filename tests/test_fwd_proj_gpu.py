"""The "LBS + projection" form of the forward (mh_lbs_forward_proj, round 4) against the stand-alone passes it replaces.

The skinning epilogue projects every vertex to NDC, reports the vertices near the previous launch's screen-box /
lowest-vertex extremes with atomics, and flags bodies whose vertices left the pixel-row band of their face lists; the
rasteriser's preparation then reads those instead of passing over the vertices, and the contact term takes the lowest
vertex from the reported key.  Everything is compared bit for bit with the passes over the stored vertices:
projected vertices, screen windows, selection keys (40 B per window pixel), lowest vertex -- over optimisation cycles
that include RMSprop's first, largest steps, a jump of several pixels (nobody reports -> the fallback scans), a fresh
workspace holding random bytes, and frames behind the camera."""
import ctypes

import numpy as np
import pytest
import torch

from test_fit_full_gpu import _setup

pytestmark = pytest.mark.gpu


def _ndc(raster, e):
    from mhhip import _lib
    off = (ctypes.c_size_t * 6)()
    _lib.lib().mh_raster_debug_offsets.argtypes = [ctypes.c_int] * 6 + [ctypes.POINTER(ctypes.c_size_t)]
    _lib.check(_lib.lib().mh_raster_debug_offsets(*raster.dims, off))
    n = e.B * e.V * 3 * 4
    return raster.ws[off[0]:off[0] + n].view(torch.int32).clone()


def _lowest(e):
    """(stand-alone kernel, from the forward's keys): low_idx, low_xyz"""
    from mhhip import _lib
    L = _lib.lib()
    st = _lib.stream_ptr(e.dev)
    i0, x0 = torch.zeros(e.B, dtype=torch.int32, device=e.dev), torch.zeros(e.B, 3, device=e.dev)
    i1, x1 = torch.zeros_like(i0), torch.zeros_like(x0)
    _lib.check(L.mh_lowest_vertex(e.verts.data_ptr(), e.B, e.V, i0.data_ptr(), x0.data_ptr(), st))
    _lib.check(L.mh_lowest_resolve(e.verts.data_ptr(), e.B, e.V, e._lowkey, i1.data_ptr(), x1.data_ptr(), st))
    torch.cuda.synchronize()
    return i0, x0, i1, x1


@pytest.mark.parametrize('T,N,W,H,batch', [(100, 4, 240, 135, 10), (12, 3, 96, 54, 3), (10, 2, 48, 80, 5), (7, 5, 64, 64, 7)])
def test_projection_epilogue_equals_the_passes_over_the_vertices(smpl_struct, smpl_regs, oracle_model, tmp_path, T, N, W, H, batch):
    from mhhip.raster import RasterTerms, set_sort_margin
    opt, dl, o, batches, seq = _setup(smpl_struct, smpl_regs, oracle_model, tmp_path, T, N, W, H, batch, 53, True)
    opt._stage_from_dataloader(dl)
    e = opt.engine
    proj, plain = RasterTerms(e), RasterTerms(e)
    proj.ws.copy_(torch.randint(0, 256, proj.ws.shape, dtype=torch.uint8, device=proj.ws.device))     # a workspace holds anything
    proj.init_workspace()
    gv, log = torch.zeros_like(e.verts), torch.zeros(16, device=e.dev)
    old = set_sort_margin(1)
    try:
        lr = 0.01
        for c in range(30):
            e.cycle(c, raster=proj)                      # forward with the projection epilogue -> preparation without a vertex pass
            torch.cuda.synchronize()
            assert e._projected_into is proj
            win1, koff1, k1 = proj.selection(e)
            ndc1 = _ndc(proj, e)
            i0, x0, i1, x1 = _lowest(e)
            assert torch.equal(i0, i1) and torch.equal(x0.view(torch.int32), x1.view(torch.int32)), 'cycle %d: lowest vertex' % c
            plain(e, gv, log, phases=1)                  # the same vertices through the stand-alone projection pass
            torch.cuda.synchronize()
            win0, koff0, k0 = plain.selection(e)
            ndc0 = _ndc(plain, e)
            assert torch.equal(ndc1, ndc0), 'cycle %d: %d projected coordinates differ' % (c, int((ndc1 != ndc0).sum()))
            assert (win1 == win0).all(), 'cycle %d: windows differ for bodies %s' % (c, np.nonzero((win1 != win0).any(axis=1))[0][:8])
            assert k1.shape == k0.shape and (k1 == k0).all(), 'cycle %d: %d window pixels differ' % (c, int((k1 != k0).any(axis=1).sum()))
            e.step(lr)
            lr *= 0.99
            if c == 11:                                  # a jump of a few pixels for every third frame, sideways for another third
                e.leaf('poses_T')[::3, :, 1] += 0.08
                e.leaf('poses_T')[1::3, :, 0] -= 0.11
            if c == 17:                                  # a frame behind the camera, one far outside the image
                e.leaf('poses_T')[0, :, 2] = -3.0
                e.leaf('poses_T')[-1, :, 0] = 40.0
        seen, rebuilt = proj.sort_counters(e)
    finally:
        set_sort_margin(old)
    assert seen == 30 * e.B
    assert e.B <= rebuilt < 0.9 * seen


def test_forward_proj_is_the_plain_forward_for_the_vertices(smpl_struct, smpl_regs, oracle_model, tmp_path):
    """verts / v_posed of the two forms are the same bits (the epilogue only adds outputs)"""
    from mhhip.raster import RasterTerms
    T, N, W, H, batch = 9, 3, 96, 54, 3
    opt, dl, o, batches, seq = _setup(smpl_struct, smpl_regs, oracle_model, tmp_path, T, N, W, H, batch, 59, True)
    opt._stage_from_dataloader(dl)
    e = opt.engine
    r = RasterTerms(e)
    e.forward(regress=False, raster=r)
    torch.cuda.synchronize()
    v1, q1 = e.verts.clone(), e.vposed.clone()
    e.forward(regress=False)
    torch.cuda.synchronize()
    assert torch.equal(v1.view(torch.int32), e.verts.view(torch.int32))
    assert torch.equal(q1.view(torch.int32), e.vposed.view(torch.int32))
