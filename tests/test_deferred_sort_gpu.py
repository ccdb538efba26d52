"""Deferred face sorts (round 6): a body whose vertices have used up HALF of the band its kept face lists
cover is sorted again beside the gradient kernel of the launch that notices it -- off the chain -- and only a jump of the
whole margin inside one cycle still sorts in the rasteriser's preparation.  The lists are an ORDER and a superset: the
selection keys must be the same bits as those of a launch that sorts afresh, and the sorts must really leave the chain."""
import numpy as np
import pytest
import torch

from test_fit_full_gpu import _setup

pytestmark = pytest.mark.gpu


def _run(smpl_struct, smpl_regs, oracle_model, tmp_path, defer, cycles=36, jump_at=20):
    from mhhip.raster import RasterTerms, set_sort_margin, set_sort_defer
    T, N, W, H, batch = 24, 3, 240, 135, 6
    old_d = set_sort_defer(defer)
    old_m = set_sort_margin(1)
    try:
        opt, dl, o, batches, seq = _setup(smpl_struct, smpl_regs, oracle_model, tmp_path, T, N, W, H, batch, 53, True)
        opt._stage_from_dataloader(dl)
        e = opt.engine
        kept, fresh = RasterTerms(e), RasterTerms(e)
        kept.ws.copy_(torch.randint(0, 256, kept.ws.shape, dtype=torch.uint8, device=kept.ws.device))     # a workspace holds anything
        kept.init_workspace()
        gv, log = torch.zeros_like(e.verts), torch.zeros(16, device=e.dev)
        lr = 0.01
        per_cycle = []
        for c in range(cycles):
            e.cycle(c, raster=kept)                      # selection on the kept lists + gradients (+ the deferred sorts)
            torch.cuda.synchronize()
            _, _, k1 = kept.selection(e)
            set_sort_margin(0)
            fresh(e, gv, log, phases=1)                  # the same vertices, lists sorted now, on a second workspace
            torch.cuda.synchronize()
            _, _, k0 = fresh.selection(e)
            set_sort_margin(1)
            assert k1.shape == k0.shape and (k1 == k0).all(), 'cycle %d (defer %.2f): %d window pixels differ' % (
                c, defer, int((k1 != k0).any(axis=1).sum()))
            per_cycle.append(kept.sort_counters3(e))
            e.step(lr)
            lr *= 0.99
            if c == jump_at:                             # a jump of several rows: must sort on the chain, at once
                e.leaf('poses_T')[::3, :, 1] += 0.08
        return e.B, per_cycle, e.params.clone()
    finally:
        set_sort_margin(old_m)
        set_sort_defer(old_d)


def test_deferred_sorts_keep_the_keys_and_leave_the_chain(smpl_struct, smpl_regs, oracle_model, tmp_path):
    B, on, p_on = _run(smpl_struct, smpl_regs, oracle_model, tmp_path, 0.5)
    B2, off, p_off = _run(smpl_struct, smpl_regs, oracle_model, tmp_path, 0.0)
    seen, rebuilt, deferred = on[-1]
    seen0, rebuilt0, deferred0 = off[-1]
    assert seen == seen0 == len(on) * B and deferred0 == 0
    chain, chain0 = rebuilt - deferred, rebuilt0
    print('sorts on the chain: %d with deferred sorts (%d more beside the gradient kernel), %d without' % (chain, deferred, chain0))
    assert deferred > 0
    assert chain >= B                                    # the first launch sorts every body where it is
    # everything beyond the first launch and the jump: RMSprop's first, largest steps (most of these 36 cycles) jump whole margins, the rest must leave the chain
    assert chain - B < 0.75 * (chain0 - B), (chain, chain0, B)
    # the jump itself is sorted on the chain in the launch that sees it
    ch = [r[1] - r[2] for r in on]
    d_chain = [ch[j] - ch[j - 1] for j in range(1, len(ch))]
    print('sorts on the chain per cycle:', [ch[0]] + d_chain)
    assert max(d_chain[20:22]) >= B // 3


def test_settings_round_trip():
    from mhhip.raster import set_sort_defer
    old = set_sort_defer(0.25)
    assert abs(set_sort_defer(old) - 0.25) < 1e-7
    from mhhip import _lib
    assert _lib.lib().mh_raster_set_sort_defer(1.5) != 0 and b'fraction' in _lib.lib().mh_last_error()
