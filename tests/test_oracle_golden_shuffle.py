"""The shipped batch semantics (``shuffle: True``, configs/predict_mupots.yml:14): the reference's ``fit`` run with
``DataLoader(shuffle=True)`` under a fixed ``torch.manual_seed`` (tests/golden/make_golden_shuffle.py ->
reference_shuffle_cpu.npz) against (a) the oracle fed with the recorded index batches and (b) the host logic of the
drop-in that re-draws those batches from the dataloader's samplers (``_cycle_batch_tables``): same seed, same batches."""
import os
import types

import numpy as np
import pytest
import torch

import golden_inputs as gi
from test_oracle_golden import LEAVES, _new_oracle, _oracle_leaves, close

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def shuf():
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'reference_shuffle_cpu.npz'), allow_pickle=False)


def batches_of(fin, table):
    out = []
    for idx in table:
        idx = np.asarray(idx)
        out.append(dict(idxs=torch.tensor(idx, dtype=torch.int64), pose2d=torch.tensor(fin['pose2d'][idx]),
                        seg_mask=torch.tensor(fin['seg_mask'][idx]), depths=torch.tensor(fin['depths'][idx]),
                        poses_smpl=torch.tensor(fin['poses_smpl'][idx])))
    return out


def _oracle(oracle_model, golden):
    fin = gi.fit_inputs()
    o = _new_oracle(oracle_model, fin, True)
    o.init_optimized_variables(fin['pose2d'], fin['poses_smpl'], fin['betas_smpl'], fin['valid_smpl'],
                               poses_T=golden['fit_init_poses_T'])
    o.update_scene_pointcloud(fin['scene_depth'], fin['scene_mask'])
    return fin, o


def test_fixture_is_a_shuffled_partition(shuf):
    b = shuf['shuf_batches']
    assert b.shape == (5, 4, 5)
    for c in range(5):
        assert sorted(b[c].reshape(-1).tolist()) == list(range(20))
    assert not (b[0] == b[1]).all() and not (np.diff(b[0], axis=1) == 1).all()


def test_first_cycle_gradients_with_shuffled_batches(golden, shuf, oracle_model):
    fin, o = _oracle(oracle_model, golden)
    o.cycle_grads(batches_of(fin, shuf['shuf_batches'][0]))
    for n, p in zip(LEAVES, o.leaves()):
        g = shuf['shuf_k1_grad_' + n]
        got = p.grad.numpy() if p.grad is not None else np.zeros_like(g)
        close(got, g, 3e-4 * max(np.abs(g).max(), 1e-6))
    # ... and the contiguous partition is a DIFFERENT problem: the foot-sliding pairs change the vertex gradients
    fin, o2 = _oracle(oracle_model, golden)
    o2.cycle_grads(batches_of(fin, np.arange(20).reshape(4, 5)))
    d = np.abs(o2.poses_smpl.grad.numpy() - shuf['shuf_k1_grad_poses_smpl'].reshape(o2.poses_smpl.shape)).max()
    assert d > 1e-3 * np.abs(shuf['shuf_k1_grad_poses_smpl']).max(), d


@pytest.mark.parametrize('k', [1, 5])
def test_fit_with_shuffled_batches(golden, shuf, oracle_model, k):
    fin, o = _oracle(oracle_model, golden)
    o.fit(lambda c: batches_of(fin, shuf['shuf_batches'][c]), k)
    got = _oracle_leaves(o)
    for n in LEAVES:
        close(got[n], shuf['shuf_k%d_%s' % (k, n)], {1: 2e-5, 5: 2e-4}[k])


class _DS(torch.utils.data.Dataset):
    def __len__(self):
        return 20

    def __getitem__(self, i):
        return dict(idxs=i)


@pytest.mark.parametrize('workers', [0, 2])
def test_drop_in_draws_the_reference_batches_from_the_samplers(shuf, workers):
    """indices only: one ``_base_seed`` draw + the sampler's own draws per cycle, like ``for data in dataloader``"""
    from mhmocap.optimizer import SMPLDepthSequenceOptimizer as Opt
    dl = torch.utils.data.DataLoader(_DS(), batch_size=int(shuf['shuf_batch']), shuffle=True, num_workers=workers)
    assert Opt._is_shuffled(dl) and not Opt._is_shuffled(torch.utils.data.DataLoader(_DS(), batch_size=5))
    fake = types.SimpleNamespace(engine=types.SimpleNamespace(nbatches=4, batch=5), num_frames=20, device='cpu')
    torch.manual_seed(int(shuf['shuf_seed']))
    tab = Opt._cycle_batch_tables(fake, dl, 5).numpy().reshape(5, 4, 5)
    np.testing.assert_array_equal(tab, shuf['shuf_batches'])
    # cross-check against the loader itself: the same seed, the real iteration
    torch.manual_seed(int(shuf['shuf_seed']))
    seen = [[int(i) for i in d['idxs']] for _ in range(2) for d in dl]
    np.testing.assert_array_equal(np.array(seen).reshape(2, 4, 5), shuf['shuf_batches'][:2])


def test_ragged_last_batch_is_padded():
    from mhmocap.optimizer import SMPLDepthSequenceOptimizer as Opt

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 13

        def __getitem__(self, i):
            return dict(idxs=i)
    dl = torch.utils.data.DataLoader(DS(), batch_size=5, shuffle=True)
    fake = types.SimpleNamespace(engine=types.SimpleNamespace(nbatches=3, batch=5), num_frames=13, device='cpu')
    tab = Opt._cycle_batch_tables(fake, dl, 2).numpy().reshape(2, 3, 5)
    assert (tab[:, 2, 3:] == -1).all() and (tab[:, :2] >= 0).all() and sorted(tab[0][tab[0] >= 0].tolist()) == list(range(13))
