"""The drop-in's ``fit`` with a frame count that is no multiple of the batch size (20 frames in batches of 6), sequential and
shuffled loaders, against the REFERENCE's own run (tests/golden/reference_ragged_cpu.npz): the weight of the shape prior, the
foot-sliding normalisation and the per-batch scale terms follow the actual size of every batch (optimizer.py:512-539)."""
import os

import numpy as np
import pytest
import torch

from test_optimizer_gpu import LEAVES, _DS, _leaf, _start

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def rag():
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'reference_ragged_cpu.npz'), allow_pickle=False)


@pytest.mark.parametrize('tag', ['seq', 'shuf'])
def test_first_cycle_gradients_ragged(golden, rag, smpl_struct, smpl_regs, tmp_path, tag):
    fin, opt = _start(smpl_struct, smpl_regs, tmp_path, golden, True)
    dl = torch.utils.data.DataLoader(_DS(fin), batch_size=int(rag['rag_batch']), shuffle=(tag == 'shuf'))
    opt._stage_from_dataloader(dl)
    e = opt.engine
    assert (e.batch, e.nbatches) == (6, 4)
    if tag == 'shuf':
        torch.manual_seed(int(rag['rag_seed']))
        tab = opt._cycle_batch_tables(dl, 1)
        t = tab.cpu().numpy().reshape(4, 6)
        np.testing.assert_array_equal(t.reshape(-1)[:20], rag['rag_shuf_order'][0])       # 6, 6, 6, 2 (+ 4 empty positions)
        assert (t[3, 2:] == -1).all()
        e.set_batch_table(tab[0])
    e.cycle(0)
    for n in LEAVES:
        g = rag['rag_%s_k1_grad_%s' % (tag, n)]
        got = _leaf(opt, n, e.grads).reshape(g.shape)
        np.testing.assert_allclose(got, g, atol=3e-4 * max(np.abs(g).max(), 1e-6), err_msg=n)


@pytest.mark.parametrize('tag,k', [('seq', 1), ('seq', 5), ('shuf', 1), ('shuf', 5)])
def test_fit_ragged_matches_reference(golden, rag, smpl_struct, smpl_regs, tmp_path, tag, k):
    fin, opt = _start(smpl_struct, smpl_regs, tmp_path, golden, True)
    torch.manual_seed(int(rag['rag_seed']))
    log = opt.fit(torch.utils.data.DataLoader(_DS(fin), batch_size=int(rag['rag_batch']), shuffle=(tag == 'shuf')), num_iter=k)
    tol = {1: 2e-5, 5: 2e-4}[k]
    for n in LEAVES:
        want = rag['rag_%s_k%d_%s' % (tag, k, n)]
        np.testing.assert_allclose(_leaf(opt, n).reshape(want.shape), want, atol=tol, err_msg=n)
    assert len(log) == k and all(l['reg_foot_sliding'] > 0 for l in log)
