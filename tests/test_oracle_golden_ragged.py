"""A frame count that is no multiple of the batch size (20 frames in batches of 6: 6, 6, 6, 2), sequential and shuffled:
the reference's own ``fit`` (tests/golden/make_golden_ragged.py -> reference_ragged_cpu.npz) against the oracle fed with the
same index batches.  What depends on the ACTUAL size of a batch in the reference: the shape prior's weight
(optimizer.py:523-525), the foot-sliding normalisation of the short batch (:512-518), the per-batch scale terms (:531-539)."""
import os

import numpy as np
import pytest
import torch

import golden_inputs as gi
from test_oracle_golden import LEAVES, _new_oracle, _oracle_leaves, close
from test_oracle_golden_shuffle import batches_of

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZES = [6, 6, 6, 2]


@pytest.fixture(scope='module')
def rag():
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'reference_ragged_cpu.npz'), allow_pickle=False)


def split(order):
    """one pass of the loader (20 frames in the order it asked for them) -> its batches"""
    edges = np.cumsum([0] + SIZES)
    return [np.asarray(order[a:b]) for a, b in zip(edges[:-1], edges[1:])]


def _oracle(oracle_model, golden):
    fin = gi.fit_inputs()
    o = _new_oracle(oracle_model, fin, True)
    o.init_optimized_variables(fin['pose2d'], fin['poses_smpl'], fin['betas_smpl'], fin['valid_smpl'],
                               poses_T=golden['fit_init_poses_T'])
    o.update_scene_pointcloud(fin['scene_depth'], fin['scene_mask'])
    return fin, o


def test_fixture_orders(rag):
    assert int(rag['rag_batch']) == 6
    assert (rag['rag_seq_order'] == np.arange(20)).all()
    for c in range(5):
        assert sorted(rag['rag_shuf_order'][c].tolist()) == list(range(20))
    assert not (rag['rag_shuf_order'][0] == rag['rag_shuf_order'][1]).all()


@pytest.mark.parametrize('tag', ['seq', 'shuf'])
def test_first_cycle_gradients_with_a_short_last_batch(golden, rag, oracle_model, tag):
    fin, o = _oracle(oracle_model, golden)
    o.cycle_grads(batches_of(fin, split(rag['rag_%s_order' % tag][0])))
    for n, p in zip(LEAVES, o.leaves()):
        g = rag['rag_%s_k1_grad_%s' % (tag, n)]
        got = p.grad.numpy() if p.grad is not None else np.zeros_like(g)
        close(got, g, 3e-4 * max(np.abs(g).max(), 1e-6))
    # ... and batches of five are a different problem for the shape prior (4 x 5 x L1 against (6 + 6 + 6 + 2) x L1 is the same
    # total, but the foot-sliding pairs and normalisations are not)
    fin, o2 = _oracle(oracle_model, golden)
    o2.cycle_grads(batches_of(fin, np.arange(20).reshape(4, 5)))
    g = rag['rag_seq_k1_grad_poses_smpl']
    assert np.abs(o2.poses_smpl.grad.numpy() - g.reshape(o2.poses_smpl.shape)).max() > 1e-4 * np.abs(g).max()


@pytest.mark.parametrize('tag', ['seq', 'shuf'])
@pytest.mark.parametrize('k', [1, 5])
def test_fit_with_a_short_last_batch(golden, rag, oracle_model, tag, k):
    fin, o = _oracle(oracle_model, golden)
    o.fit(lambda c: batches_of(fin, split(rag['rag_%s_order' % tag][c])), k)
    got = _oracle_leaves(o)
    for n in LEAVES:
        close(got[n], rag['rag_%s_k%d_%s' % (tag, k, n)], {1: 2e-5, 5: 2e-4}[k])
