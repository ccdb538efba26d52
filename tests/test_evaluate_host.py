"""Host logic of the overlay's evaluator (``mhmocap/evaluate.py``) against the reference's own run
(tests/golden/reference_eval_cpu.npz, tests/golden/make_golden_eval.py): with the reference body model's sparse joints
handed in through a stand-in ``SMPLPY`` everything else -- layout maps, projection, matching, the compacted rows, the
jitter -- must reproduce the reference's float32 outputs bit for bit.  The device leg (the drop-in SMPL as ``SMPLPY``) is
tests/test_evaluate_gpu.py."""
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = ['mupots', 'dist', 'panoptic']
OUT = ['abs_dist', 'rel_dist', 'valid_joints', 'abs_root_pos_err', 'valid_root', 'abs_jitter']


@pytest.fixture(scope='module')
def golden_eval():
    return dict(np.load(os.path.join(HERE, 'golden', 'reference_eval_cpu.npz')))


def case_inputs(g, tag):
    ov = {k: g['%s_ov_%s' % (tag, k)] for k in ['poses_T', 'poses_smpl', 'betas_smpl', 'scale_factor', 'valid_smpl']}
    kd = g[tag + '_kd']
    return ov, g[tag + '_ref_poses3d'], g[tag + '_visibility'], (None if kd.size == 0 else kd)


class _Recorded(object):
    """the reference body model's answers from the fixture, behind SMPL.__call__'s keyword surface"""

    def __init__(self, g, tag):
        self.g, self.tag, self.calls = g, tag, 0

    def __call__(self, betas=None, poses=None, **kw):
        self.calls += 1
        assert betas.shape[1] == 10 and poses.shape[1] == 72 and betas.shape[0] == poses.shape[0]
        return {k: torch.tensor(self.g['%s_%s' % (self.tag, k)]) for k in ['joints_mupots', 'joints_alphapose']}


@pytest.mark.parametrize('tag', CASES)
def test_outputs_of_the_reference_evaluator(golden_eval, tag):
    from mhmocap import evaluate as ev
    g = golden_eval
    ov, gt, vis, kd = case_inputs(g, tag)
    body = _Recorded(g, tag)
    keep = (gt.copy(), vis.copy())
    m = ev.compute_smpl_pred_error_3dproj(ov, gt, vis, body, g['cam_K'], Kd=kd)
    assert body.calls == 1, 'all T*N bodies go through the body model in one call'
    assert np.array_equal(gt, keep[0]) and np.array_equal(vis, keep[1])
    assert sorted(m) == sorted(OUT)
    for k in OUT:
        want = g['%s_out_%s' % (tag, k)]
        assert m[k].dtype == np.float32 and m[k].shape == want.shape, k
        if k.startswith('valid'):
            assert np.array_equal(m[k], want), k           # which pairs were matched, which joints count
        else:
            assert np.array_equal(m[k], want), '%s: %.3e' % (k, np.abs(m[k] - want).max())      # same float32 operations in the same order
    s = [ev.masked_average_error(m['abs_dist'], m['valid_joints']), ev.masked_average_error(m['rel_dist'], m['valid_joints']),
         ev.masked_average_error(m['abs_root_pos_err'], m['valid_root']), ev.masked_average_pck(m['rel_dist'], m['valid_joints'], 0.15),
         ev.masked_average_pck(m['abs_root_pos_err'], m['valid_root'], 0.25), ev.masked_average_error(m['abs_jitter'], m['valid_joints'])]
    np.testing.assert_allclose(s, g[tag + '_summary'], rtol=1e-6)


def test_more_reference_persons_than_predictions_compacts_the_rows(golden_eval):
    """K = 3, N = 2: a frame has two matched pairs, they fill rows 0 and 1 whichever reference persons they are
    (evaluate.py:259: the row index enumerates the assignment), row 2 stays zero"""
    g = golden_eval
    for k in ['abs_dist', 'valid_joints', 'abs_root_pos_err']:
        assert not g['mupots_out_' + k][:, 2].any()
    assert g['mupots_out_valid_joints'][:, :2].any()


def test_layout_maps():
    from mhmocap import evaluate as ev
    x = np.arange(2 * 19 * 3, dtype=np.float64).reshape(2, 19, 3)
    y = ev.map_cmu_panoptic_to_mupots15j(x)
    assert y.dtype == np.float32 and y.shape == (2, 15, 3)
    assert np.array_equal(y[:, 0], x[:, 1]) and np.array_equal(y[:, 14], x[:, 2]) and np.array_equal(y[:, 8], x[:, 12])
    a = np.random.RandomState(0).randn(3, 17, 3).astype(np.float32)
    b = ev.map_alphapose_to_mupots15j(a)
    assert np.array_equal(b[:, 1], np.float32(0.5) * a[:, 5] + np.float32(0.5) * a[:, 6])
    assert np.array_equal(b[:, 14], np.float32(0.5) * a[:, 11] + np.float32(0.5) * a[:, 12]) and np.array_equal(b[:, 10], a[:, 16])
    with pytest.raises(AssertionError):
        ev.map_alphapose_to_mupots15j(a[0])


def test_other_joint_counts_are_refused(golden_eval):
    from mhmocap import evaluate as ev
    g = golden_eval
    ov, gt, vis, kd = case_inputs(g, 'mupots')
    with pytest.raises(AssertionError, match='only 17'):
        ev.compute_smpl_pred_error_3dproj(ov, gt[:, :, :16], vis[:, :, :16], _Recorded(g, 'mupots'), g['cam_K'])


def test_masked_averages():
    from mhmocap import evaluate as ev
    d = np.array([[0.1, 0.2], [0.3, 10.0]])
    v = np.array([[1.0, 0.6], [0.5, 0.0]])                # 0.5 does not count
    assert ev.masked_average_error(d, v) == pytest.approx(0.15, rel=1e-6)
    assert ev.masked_average_pck(d, v, 0.15) == pytest.approx(0.5)
    assert ev.masked_average_error(d, np.zeros_like(v)) == 0.0
    with pytest.raises(AssertionError):
        ev.masked_average_error(d, v[0])
