"""The selection kernel has two ways of working through a round of 64 candidate faces: small pixel boxes go through the depth
cull and a compacted pair list, large ones -- and whole rounds whose list would overflow -- through an even split of all pairs
over the lanes.  Which path a (face, pixel) pair takes depends on its round's company (the sort, the tile, the other faces),
so the keys must not depend on it: the pair arithmetic is the same inlined function in both loops, and until round 4 hipcc
fused its multiply-adds differently in the two copies -- the same pair came out with a depth one ulp apart (47 753 of 1.2 M
window pixels of the C3 launch changed a last bit when rounds were moved from one path to the other).  The arithmetic is
spelled out now (csrc/mh_raster.hip, r_eval_fast); this test holds the two paths against each other, bit for bit, through
the test aid mh_raster_set_path."""
import numpy as np
import pytest
import torch

from test_fit_full_gpu import _setup

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('T,N,W,H,batch', [(40, 4, 240, 135, 10), (10, 2, 96, 54, 5), (6, 2, 480, 270, 3)])
def test_both_paths_of_the_selection_kernel_give_the_same_keys(smpl_struct, smpl_regs, oracle_model, tmp_path, T, N, W, H, batch):
    from mhhip import _lib
    from mhhip.raster import RasterTerms
    opt, dl, o, batches, seq = _setup(smpl_struct, smpl_regs, oracle_model, tmp_path, T, N, W, H, batch, 53, True)
    opt._stage_from_dataloader(dl)
    e = opt.engine
    L = _lib.lib()
    r = RasterTerms(e)
    gv, log = torch.zeros_like(e.verts), torch.zeros(16, device=e.dev)
    e.cycle(0, raster=r)                                   # forward + both halves of the rasteriser, usual paths
    torch.cuda.synchronize()
    _, _, k_mixed = r.selection(e)
    assert L.mh_raster_get_path() == 0
    try:
        L.mh_raster_set_path(1)                            # every round through the even split
        r(e, gv, log, phases=1)
        torch.cuda.synchronize()
        _, _, k_even = r.selection(e)
    finally:
        L.mh_raster_set_path(0)
    live = int((k_mixed[:, 0] != np.uint64(0xffffffffffffffff)).sum())
    assert live > 50 * e.B                                  # the bodies are on screen
    diff = (k_mixed != k_even).any(axis=1)
    assert not diff.any(), '%d of %d window pixels differ between the two paths' % (int(diff.sum()), len(diff))
    r(e, gv, log, phases=1)                                 # and back: the switch is read per launch
    torch.cuda.synchronize()
    assert (r.selection(e)[2] == k_mixed).all()


def test_the_closing_job_gives_the_same_cycle_wherever_it_runs(smpl_struct, smpl_regs, oracle_model, tmp_path, monkeypatch):
    """The rasterised terms' closing job (per-body values, depth-range gradients, the two log sums) runs as one workgroup of the
    LBS backward's pose kernel (mh_raster_terms_deferred + mh_lbs_backward_kp_fin) or as k_raster_finish's own launch
    (MHHIP_NO_DEFER=1): every gradient leaf and the log row of a cycle are the same bits either way (deterministic scatter)."""
    from mhhip.raster import RasterTerms, set_deterministic
    opt, dl, o, batches, seq = _setup(smpl_struct, smpl_regs, oracle_model, tmp_path, 30, 3, 240, 135, 10, 59, True)
    opt._stage_from_dataloader(dl)
    e = opt.engine
    r = RasterTerms(e)
    old = set_deterministic(True)
    try:
        out = []
        for no_defer in ('0', '1', '0'):
            monkeypatch.setenv('MHHIP_NO_DEFER', no_defer)
            e.cycle(0, raster=r)
            torch.cuda.synchronize()
            out.append((e.grads.clone(), e.log[0].clone(), e.depth_body.clone(), e.sil_body.clone()))
    finally:
        set_deterministic(old)
    assert float(out[0][0].abs().max()) > 0 and float(out[0][2].abs().max()) > 0
    zmin = e.leaf('zmin_lin', out[0][0])
    assert float(zmin.abs().max()) > 0                      # the depth-range leaves got their gradient from the job
    for a, b in ((out[0], out[1]), (out[0], out[2])):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
