"""The log row of a replayed cycle travels to its place in the launch of the update (``mh_rmsprop_step_log``; one small
launch less behind every graph replay): the kernel against update + copy, and the engine's bookkeeping -- whoever needs the
row first (the next ``step``, ``read_log``, the next cycle, eager or replayed) gets it there, exactly once."""
import numpy as np
import pytest
import torch

import test_fit_full_gpu as tf

pytestmark = pytest.mark.gpu


def test_update_with_log_copy_is_update_plus_copy():
    from mhhip import engine
    rng = np.random.RandomState(4)
    for n in (1, 255, 60444):
        p = torch.tensor(rng.normal(0, 1, n).astype(np.float32), device='cuda:0')
        g = torch.tensor(rng.normal(0, 1, n).astype(np.float32), device='cuda:0')
        sq = torch.tensor(rng.uniform(0, 1, n).astype(np.float32), device='cuda:0')
        buf = torch.tensor(rng.normal(0, 1, n).astype(np.float32), device='cuda:0')
        p2, sq2, buf2 = p.clone(), sq.clone(), buf.clone()
        src = torch.arange(16, dtype=torch.float32, device='cuda:0') + 0.5
        log = torch.full((3, 16), -1.0, device='cuda:0')
        engine.rmsprop_step(p, g, sq, buf, 0.0093)
        engine.rmsprop_step_log(p2, g, sq2, buf2, 0.0093, src, log[1])
        assert torch.equal(p, p2) and torch.equal(sq, sq2) and torch.equal(buf, buf2)
        assert torch.equal(log[1], src) and bool((log[0] == -1).all()) and bool((log[2] == -1).all())


def test_null_log_pointers_are_refused():
    from mhhip import _lib
    L = _lib.lib()
    x = torch.zeros(8, device='cuda:0')
    rc = L.mh_rmsprop_step_log(_lib.ptr(x), _lib.ptr(x), _lib.ptr(x), _lib.ptr(x), 8, 0.01, 0.5, 0.9, 1e-8, None, None, 4, None)
    assert rc != 0 and b'log' in L.mh_last_error()
    assert L.mh_rmsprop_step_log(_lib.ptr(x), _lib.ptr(x), _lib.ptr(x), _lib.ptr(x), 8, 0.01, 0.5, 0.9, 1e-8, None, None, 0, None) == 0
    torch.cuda.synchronize()


def test_every_reader_finds_the_row(smpl_struct, smpl_regs, oracle_model, tmp_path):
    """replayed cycles with and without a step in between, an eager cycle in between, read_log first: the rows equal those
    of eager cycles from the same leaves (eager cycles write their row themselves)"""
    from mhhip.raster import RasterTerms
    opt, dl, o, batches, seq = tf._setup(smpl_struct, smpl_regs, oracle_model, tmp_path, 6, 2, 96, 54, 4, 77, False)
    opt._stage_from_dataloader(dl)
    e = opt.engine
    raster = RasterTerms(e)
    e.log.zero_()
    e.cycle(0, raster=raster)                       # eager: row 0
    want = e.read_log(1)[0]
    e.cycle_graphed(1, raster=raster)               # captured on first use; row 1 pending
    assert e._log_pending == 1
    e.cycle_graphed(2, raster=raster)               # no step in between: row 1 flushed before the replay overwrites the staging row
    assert e._log_pending == 2
    rows = e.read_log(3)                            # read first: flushes row 2
    assert e._log_pending is None
    for r in rows[1:]:
        for k in want:
            np.testing.assert_allclose(r[k], want[k], rtol=1e-5, atol=1e-9, err_msg=k)
    e.cycle_graphed(3, raster=raster)
    e.cycle(4, raster=raster)                       # an eager cycle clears the staging row: row 3 must be out before that
    assert e._log_pending is None
    e.cycle_graphed(5, raster=raster)
    before = e.params.clone()
    e.step(0.01)                                    # row 5 rides with the update
    assert e._log_pending is None and not torch.equal(before, e.params)
    rows = e.read_log(6)
    for i in (3, 4, 5):
        for k in want:
            np.testing.assert_allclose(rows[i][k], want[k], rtol=1e-5, atol=1e-9, err_msg='row %d %s' % (i, k))
