"""The drop-in on the smallest problems -- one frame (no temporal pair anywhere), one human (7 frames in batches of 3: a batch
of a single frame), two frames in a batch larger than the sequence -- against the REFERENCE's own warm-up and ``fit``
(tests/golden/reference_edge_cpu.npz; the oracle is pinned to the same fixture in tests/test_oracle_golden_edge.py)."""
import os

import numpy as np
import pytest
import torch

from edge_inputs import VARIANTS, sub_inputs
from test_optimizer_gpu import LEAVES, _DS, _leaf
from test_round2_gaps_gpu import _new_opt

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def edge():
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'reference_edge_cpu.npz'), allow_pickle=False)


@pytest.mark.parametrize('tag', sorted(VARIANTS))
def test_warm_up_on_the_smallest_problems(edge, smpl_struct, smpl_regs, tmp_path, tag):
    f, _ = sub_inputs(tag)
    opt = _new_opt(smpl_struct, smpl_regs, tmp_path, f)
    log = opt.init_optimized_variables(f['pose2d'], f['poses_smpl'], f['betas_smpl'], f['valid_smpl'], num_iter=5)
    np.testing.assert_allclose([float(l['loss_2d']) for l in log], edge['edge_%s_init_log' % tag], rtol=1e-4)
    for n in LEAVES:
        want = edge['edge_%s_init_%s' % (tag, n)]
        err = np.abs(_leaf(opt, n).reshape(want.shape) - want)
        assert err.max() <= (5e-3 if n in ('poses_T', 'zmax_lin') else 2e-5), (n, float(err.max()))


@pytest.mark.parametrize('tag', sorted(VARIANTS))
def test_cycle_and_fits_on_the_smallest_problems(edge, smpl_struct, smpl_regs, tmp_path, tag):
    f, b = sub_inputs(tag)

    def start(sub):
        d = tmp_path / sub
        d.mkdir()
        opt = _new_opt(smpl_struct, smpl_regs, d, f)
        opt.init_optimized_variables(f['pose2d'], f['poses_smpl'], f['betas_smpl'], f['valid_smpl'], num_iter=0)
        e = opt.engine
        e.leaf('poses_T').copy_(torch.tensor(edge['edge_%s_init_poses_T' % tag]).view(f['T'], f['N'], 3))
        e.leaf('zmax_lin').copy_(torch.tensor(edge['edge_%s_init_zmax_lin' % tag]).view(-1))
        opt.scene_depth = f['scene_depth']
        opt.update_scene_pointcloud(f['scene_depth'], f['scene_mask'])
        return opt

    opt = start('g')
    opt._stage_from_dataloader(torch.utils.data.DataLoader(_DS(f), batch_size=b, shuffle=False))
    opt.engine.cycle(0)
    for n in LEAVES:
        g = edge['edge_%s_k1_grad_%s' % (tag, n)]
        np.testing.assert_allclose(_leaf(opt, n, opt.engine.grads).reshape(g.shape), g, atol=3e-4 * max(np.abs(g).max(), 1e-6), err_msg=n)
    for k in (1, 3):
        opt = start('f%d' % k)
        log = opt.fit(torch.utils.data.DataLoader(_DS(f), batch_size=b, shuffle=False), num_iter=k)
        assert len(log) == k
        for n in LEAVES:
            want = edge['edge_%s_k%d_%s' % (tag, k, n)]
            err = np.abs(_leaf(opt, n).reshape(want.shape) - want)
            tol = {1: 2e-5, 3: 1e-4}[k]
            assert (err > tol).mean() <= 0.005 and err.max() <= 1e-3, '%s after %d: %.4f above %g, max %.2e' % (
                n, k, float((err > tol).mean()), tol, float(err.max()))
