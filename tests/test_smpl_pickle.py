"""``SMPL_{GENDER}.pkl`` plumbing (reference smpl.py:179-188): the official file stores chumpy objects and a scipy
sparse joint regressor; chumpy is not installable here, so the drop-in unpickles with a stand-in class
(mhhip/engine.py:load_smpl_pickle).  A synthetic pickle with exactly that structure (a ``chumpy.ch.Ch`` class
pickled from a throw-away module, latin1 / protocol 2 like the original) goes through the loader on the CPU, and
through ``mhmocap.smpl.SMPL(model_path=dir)`` on the GPU."""
import os
import pickle
import sys
import types

import numpy as np
import pytest
import scipy.sparse

from mhhip import synthetic


def _write_chumpy_style_pickle(path, st):
    """The pickle is written while a fake ``chumpy`` package exists, then the package is removed again."""
    pkg, mod = types.ModuleType('chumpy'), types.ModuleType('chumpy.ch')

    class Ch(object):
        def __init__(self, x):
            self.x = np.asarray(x)
            self._dirty_vars = set()
            self._itr = None

    Ch.__module__, Ch.__qualname__ = 'chumpy.ch', 'Ch'
    mod.Ch = Ch
    pkg.ch = mod
    sys.modules['chumpy'], sys.modules['chumpy.ch'] = pkg, mod
    try:
        d = {'v_template': Ch(st.v_template), 'shapedirs': Ch(st.shapedirs), 'posedirs': Ch(st.posedirs),
             'J_regressor': scipy.sparse.csc_matrix(st.J_regressor), 'weights': Ch(st.weights),
             'kintree_table': st.kintree_table, 'f': st.f, 'bs_style': 'lbs', 'bs_type': 'lrotmin',
             'J': Ch(st.J_regressor @ st.v_template)}
        with open(path, 'wb') as f:
            pickle.dump(d, f, protocol=2)
    finally:
        del sys.modules['chumpy'], sys.modules['chumpy.ch']


def test_loader_reads_chumpy_objects_without_chumpy(tmp_path, smpl_struct):
    from mhhip import engine
    path = str(tmp_path / 'SMPL_NEUTRAL.pkl')
    _write_chumpy_style_pickle(path, smpl_struct)
    assert 'chumpy' not in sys.modules
    with pytest.raises(ModuleNotFoundError):
        pickle.load(open(path, 'rb'), encoding='latin1')            # what the reference's plain unpickling does here
    d = engine.load_smpl_pickle(path)
    for k in ['v_template', 'shapedirs', 'posedirs', 'weights']:
        np.testing.assert_array_equal(np.asarray(d[k]), getattr(smpl_struct, k))
        np.testing.assert_array_equal(engine._dense(d[k]), getattr(smpl_struct, k).astype(np.float32))
    np.testing.assert_array_equal(engine._dense(d['J_regressor']), smpl_struct.J_regressor.astype(np.float32))
    np.testing.assert_array_equal(np.asarray(d['kintree_table']), smpl_struct.kintree_table)
    np.testing.assert_array_equal(np.asarray(d['f']), smpl_struct.f)


@pytest.mark.gpu
def test_smpl_from_a_pickle_directory(tmp_path, smpl_struct, smpl_regs):
    import torch
    from mhmocap.smpl import SMPL
    import golden_inputs as gi
    _write_chumpy_style_pickle(str(tmp_path / 'SMPL_NEUTRAL.pkl'), smpl_struct)
    np.save(str(tmp_path / 'a.npy'), smpl_regs['alphapose'])
    betas, poses = gi.lbs_inputs()
    a = SMPL(str(tmp_path), J_reg_alphapose_path=str(tmp_path / 'a.npy')).to('cuda:0')(betas=betas, poses=poses)
    b = SMPL(None, J_reg_alphapose_path=str(tmp_path / 'a.npy'), data_struct=smpl_struct).to('cuda:0')(betas=betas, poses=poses)
    for k in ['verts', 'joints_smpl24', 'joints_alphapose', 'j3d']:
        np.testing.assert_array_equal(a[k].detach().cpu().numpy(), b[k].detach().cpu().numpy())
