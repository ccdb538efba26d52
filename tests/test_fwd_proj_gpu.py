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
            assert proj.last_projected == 1 and e._projected_into is None     # used by the preparation, and used up (one-shot)
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


def test_halo_forward_gives_the_owners_bits(smpl_struct, smpl_regs, oracle_model, tmp_path):
    """the frame-sharded cycle skins its neighbours' boundary frames itself (``SequenceEngine._halo_forward``: the
    plain, non-FULL instantiation of the skinning kernel over N or 2N bodies with its own workspace) and relies on the result
    being the OWNER's vertices bit for bit (the owner runs the projecting, FULL instantiation over all its bodies).  One
    process is enough to hold the two instantiations against each other: the engine's own first and last frame as "halo"."""
    from mhhip.raster import RasterTerms
    T, N, W, H, batch = 32, 4, 96, 54, 4                    # 128 bodies: the owner's launch is the FULL form
    opt, dl, o, batches, seq = _setup(smpl_struct, smpl_regs, oracle_model, tmp_path, T, N, W, H, batch, 71, True)
    opt._stage_from_dataloader(dl)
    e = opt.engine
    r = RasterTerms(e)
    gen = torch.Generator(device=e.dev); gen.manual_seed(3)
    e.leaf('poses_smpl').add_(torch.randn(e.leaf('poses_smpl').shape, device=e.dev, generator=gen) * 0.2)
    e.leaf('betas').add_(torch.randn(e.leaf('betas').shape, device=e.dev, generator=gen) * 0.5)
    e.leaf('xscale').add_(0.3)
    e.forward(regress=False, raster=r)                       # the owner: projection epilogue, all bodies
    torch.cuda.synchronize()
    owner = e.verts.view(T, N, -1, 3)
    pT, ps = e.leaf('poses_T').view(T, N, 3), e.leaf('poses_smpl').view(T, N, 72)
    for frames in ([0], [T - 1], [T - 1, 0]):                # a first rank, a last rank, a middle rank (previous | next)
        h = dict(poses=torch.cat([ps[f] for f in frames]).contiguous(), transl=torch.cat([pT[f] for f in frames]).contiguous(),
                 has_prev=True, has_next=len(frames) == 2)
        e._halo_forward(h, torch.cuda.current_stream(e.dev).cuda_stream)
        torch.cuda.synchronize()
        got = [h['v_prev']] + ([h['v_next']] if len(frames) == 2 else [])
        for f, v in zip(frames, got):
            assert torch.equal(v.view(torch.int32), owner[f].view(torch.int32)), 'halo of frame %d differs from the owner\'s vertices' % f


@pytest.mark.parametrize('T,N,W,H,batch', [(50, 4, 240, 135, 10), (16, 4, 96, 54, 4), (7, 5, 64, 64, 7), (3, 1, 96, 54, 3)])
def test_pipelined_forward_gives_the_same_bits(smpl_struct, smpl_regs, oracle_model, tmp_path, T, N, W, H, batch):
    """k_skin_fwd16p (round 6: a wave issues tile i's epilogue between the matrix instructions of tile i + 1) against
    k_skin_fwd16 (tile after tile): vertices, rest-pose vertices, projected vertices, the bodies' report slots and motion flags
    -- the same bits, on full groups of 32 bodies, a ragged last group and a single small group."""
    from mhhip import _lib
    from mhhip.raster import RasterTerms
    L = _lib.lib()
    opt, dl, o, batches, seq = _setup(smpl_struct, smpl_regs, oracle_model, tmp_path, T, N, W, H, batch, 59, True)
    opt._stage_from_dataloader(dl)
    e = opt.engine
    raster = RasterTerms(e)
    e.cycle(0, raster=raster)                 # face lists sorted, previous-launch report slots filled
    e.step(0.01)
    off = (ctypes.c_size_t * 6)()
    L.mh_raster_debug_offsets.argtypes = [ctypes.c_int] * 6 + [ctypes.POINTER(ctypes.c_size_t)]
    _lib.check(L.mh_raster_debug_offsets(*raster.dims, off))
    ws0 = raster.ws.clone()
    old = L.mh_lbs_get_forward_pipeline()
    outs = []
    try:
        for pipe in (0, 1, 0, 2):
            _lib.check(L.mh_lbs_set_forward_pipeline(pipe))
            raster.ws.copy_(ws0)              # the same previous-launch slots for both
            e.verts.fill_(7.0); e.vposed.fill_(7.0)
            e.forward(regress=False, raster=raster)
            torch.cuda.synchronize()
            outs.append((e.verts.clone(), e.vposed.clone(), raster.ws[off[0]:].clone()))
    finally:
        L.mh_lbs_set_forward_pipeline(old)
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a.view(torch.uint8).view(-1), b.view(torch.uint8).view(-1))
    for a, b in zip(outs[0], outs[2]):
        assert torch.equal(a.view(torch.uint8).view(-1), b.view(torch.uint8).view(-1))
    # ... and the form with producer and consumer waves (k_skin_fwd16pc: the accumulators travel through an LDS slot)
    for a, b in zip(outs[0], outs[3]):
        assert torch.equal(a.view(torch.uint8).view(-1), b.view(torch.uint8).view(-1))
