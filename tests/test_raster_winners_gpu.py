"""Winners first (round 5, csrc/mh_raster.hip: r_face_sort / k_raster_strip).

When a body's face lists are sorted, the faces that held one of the five keys of some window pixel in the previous launch go
into a list of their own; every tile of the selection kernel rasterises that list first, so that the depth cull meets nearly
final 4th keys for the other 80 % of the faces.  It is an ORDER: the 40-byte key record of every window pixel must be the same
bits with and without it -- with stale winners (lists kept while the bodies move), winners taken from keys of random bytes,
after jumps, and for a body that leaves the image and comes back."""
import ctypes

import numpy as np
import pytest
import torch

from test_fit_full_gpu import _setup

pytestmark = pytest.mark.gpu


def _views(raster, e):
    from mhhip import _lib
    off = (ctypes.c_size_t * 3)()
    f = _lib.lib().mh_raster_debug_offsets2
    f.argtypes = [ctypes.c_int] * 6 + [ctypes.POINTER(ctypes.c_size_t)]
    _lib.check(f(*raster.dims, off))
    koff = (ctypes.c_size_t * 3)()
    _lib.check(_lib.lib().mh_raster_workspace_offsets(*raster.dims, koff))
    B = e.B
    kvalid = raster.ws[off[0]:off[0] + B * 4].view(torch.int32)
    wstate = raster.ws[off[1]:off[1] + B * 4].view(torch.int32)
    tags = raster.ws[off[2]:off[2] + B * 8].view(torch.int64)
    keys = raster.ws[koff[2]:koff[2] + B * e.H * e.W * 40].view(torch.int64)
    return kvalid, wstate, tags, keys


@pytest.mark.parametrize('T,N,W,H,batch', [(20, 4, 240, 135, 10), (10, 2, 96, 54, 5), (4, 2, 480, 270, 2)])
def test_selection_is_the_same_bits_with_and_without_the_winners_list(smpl_struct, smpl_regs, oracle_model, tmp_path, T, N, W, H, batch):
    from mhhip import _lib
    from mhhip.raster import RasterTerms, set_winners
    opt, dl, o, batches, seq = _setup(smpl_struct, smpl_regs, oracle_model, tmp_path, T, N, W, H, batch, 61, True)
    opt._stage_from_dataloader(dl)
    e = opt.engine
    L = _lib.lib()
    assert L.mh_raster_get_winners() == 1                    # the default
    r_win = RasterTerms(e)
    old = set_winners(False)
    try:
        r_ref = RasterTerms(e)
    finally:
        set_winners(old)
    kvalid, wstate, tags, keys = _views(r_win, e)
    gv, log = torch.zeros_like(e.verts), torch.zeros(16, device=e.dev)
    gen = torch.Generator(device=e.dev); gen.manual_seed(7)
    pT = e.leaf('poses_T')
    pT0 = pT.clone()
    with_winners = 0
    for c in range(12):
        if c in (3, 4, 9):
            pT.add_(torch.randn(pT.shape, device=e.dev, generator=gen) * 0.002)       # optimiser-sized steps: lists (and winners) kept
        if c == 5:
            pT[:, 0, 0] += 0.08                                                       # a jump of a few pixels: re-sort, fresh winners
        if c == 6:
            pT[:, -1, 0] += 30.0                                                      # one person out of the image ...
        if c == 7:
            pT.copy_(pT0)                                                             # ... and back
        if c == 8:
            e.leaf('poses_smpl').add_(torch.randn(e.leaf('poses_smpl').shape, device=e.dev, generator=gen) * 0.02)
        if c == 10:
            # winners from keys of random bytes: every body sorts again (tags cleared) and reads garbage face ids
            keys.copy_(torch.randint(-2 ** 62, 2 ** 62, keys.shape, device=e.dev, dtype=torch.int64, generator=gen))
            tags.zero_(); kvalid.fill_(1)
        e.forward(regress=False, raster=r_win)               # projects into r_win's workspace
        r_win(e, gv, log, phases=1)
        torch.cuda.synchronize()
        with_winners = max(with_winners, int(wstate.sum()))
        old = set_winners(False)
        try:
            e._projected_into = None
            r_ref(e, gv, log, phases=1)                      # from the vertices, no winners' list
            torch.cuda.synchronize()
        finally:
            set_winners(old)
        wa, ka, a = r_win.selection(e)
        wb, kb, b = r_ref.selection(e)
        assert np.array_equal(wa, wb) and np.array_equal(ka, kb)
        diff = (a != b).any(axis=1)
        assert not diff.any(), 'launch %d: %d of %d window pixels differ with / without the winners\' list' % (c, int(diff.sum()), len(diff))
        print('launch %2d: %7d live pixels, bodies whose lists carry winners %d of %d' % (c, int((a[:, 0] != np.uint64(0xffffffffffffffff)).sum()), int(wstate.sum()), e.B))
    assert with_winners >= e.B - N * 2                       # from the second launch on every body on screen has its winners' list


def test_winners_cull_pairs_and_the_cycle_gradients_do_not_change(smpl_struct, smpl_regs, oracle_model, tmp_path):
    """(a) the point of the list: the launch evaluates clearly fewer (face, pixel) pairs; (b) a whole cycle -- deterministic
    scatter -- gives the same gradient bits with and without it."""
    from mhhip import _lib
    from mhhip.raster import RasterTerms, set_deterministic, set_winners
    opt, dl, o, batches, seq = _setup(smpl_struct, smpl_regs, oracle_model, tmp_path, 20, 4, 240, 135, 10, 67, True)
    opt._stage_from_dataloader(dl)
    e = opt.engine
    L = _lib.lib()
    old_det = set_deterministic(True)
    L.mh_profile_enable(2)
    out = {}
    try:
        for on in (True, False):
            old = set_winners(on)
            try:
                r = RasterTerms(e)
                for c in range(3):                           # launch 0 sorts without keys, launch 1 with them
                    e.cycle(0, raster=r)
                torch.cuda.synchronize()
                p0 = r.pair_counters(e)
                e.cycle(0, raster=r)
                torch.cuda.synchronize()
                p1 = r.pair_counters(e)
            finally:
                set_winners(old)
            out[on] = (e.grads.clone(), e.log[0].clone(), (p1[2] - p0[2]) / max(1, p1[0] - p0[0]))
    finally:
        L.mh_profile_enable(0)
        set_deterministic(old_det)
    assert torch.equal(out[True][0], out[False][0]) and torch.equal(out[True][1], out[False][1])
    print('pairs evaluated per launch: winners first %.0f, without %.0f' % (out[True][2], out[False][2]))
    assert out[True][2] < 0.9 * out[False][2]
