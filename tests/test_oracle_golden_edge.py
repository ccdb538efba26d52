"""The smallest problems -- one frame, one human (with a batch of a single frame), two frames in a batch larger than the
sequence -- the reference's own warm-up and ``fit`` (tests/golden/make_golden_edge.py -> reference_edge_cpu.npz) against the
oracle: empty temporal sums, single-frame batches, normalisations by counts that can be zero."""
import os

import numpy as np
import pytest
import torch

import golden_inputs as gi
from edge_inputs import VARIANTS, batches, sub_inputs
from oracle import fit_oracle as fo
from test_oracle_golden import LEAVES, _oracle_leaves, close

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def edge():
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'reference_edge_cpu.npz'), allow_pickle=False)


def _oracle(oracle_model, f):
    stub = lambda v: (-torch.ones(v.shape[0], f['H'], f['W']) + 0.0 * v.sum(), torch.zeros(v.shape[0], f['H'], f['W']) + 0.0 * v.sum())
    return fo.SequenceOracle(oracle_model, (f['W'], f['H']), f['T'], f['cam_K'], coefs=gi.COEFS, rasteriser=stub)


@pytest.mark.parametrize('tag', sorted(VARIANTS))
def test_warm_up_on_the_smallest_problems(edge, oracle_model, tag):
    f, _ = sub_inputs(tag)
    o = _oracle(oracle_model, f)
    log = o.init_optimized_variables(f['pose2d'], f['poses_smpl'], f['betas_smpl'], f['valid_smpl'], num_iter=5)
    np.testing.assert_allclose(log, edge['edge_%s_init_log' % tag], rtol=2e-5)
    got = _oracle_leaves(o)
    for n in LEAVES:
        want = edge['edge_%s_init_%s' % (tag, n)]
        err = np.abs(got[n].reshape(want.shape) - want)
        assert err.max() <= (5e-3 if n in ('poses_T', 'zmax_lin') else 2e-5), (n, float(err.max()))


@pytest.mark.parametrize('tag', sorted(VARIANTS))
def test_cycle_and_fits_on_the_smallest_problems(edge, oracle_model, tag):
    f, b = sub_inputs(tag)

    def start():
        o = _oracle(oracle_model, f)
        o.init_optimized_variables(f['pose2d'], f['poses_smpl'], f['betas_smpl'], f['valid_smpl'],
                                   poses_T=edge['edge_%s_init_poses_T' % tag])
        o.update_scene_pointcloud(f['scene_depth'], f['scene_mask'])
        return o

    o = start()
    o.cycle_grads(batches(f, b))
    for n, p in zip(LEAVES, o.leaves()):
        g = edge['edge_%s_k1_grad_%s' % (tag, n)]
        got = p.grad.numpy() if p.grad is not None else np.zeros_like(g)
        close(got.reshape(g.shape), g, 3e-4 * max(np.abs(g).max(), 1e-6))
    for k in (1, 3):
        o = start()
        o.fit(batches(f, b), k)
        got = _oracle_leaves(o)
        for n in LEAVES:
            want = edge['edge_%s_k%d_%s' % (tag, k, n)]
            close(got[n].reshape(want.shape), want, {1: 2e-5, 3: 1e-4}[k])
