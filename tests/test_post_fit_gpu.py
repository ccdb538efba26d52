"""What the callers read after ``fit`` (predict.py:344-347): ``get_optimized_variables()`` and ``predict()`` of the drop-in
against the REFERENCE's own objects after its warm-up and three cycles (tests/golden/make_golden_post.py ->
reference_post_cpu.npz).  ``get_filtered_vertices_by_smpl()`` raises in the reference as shipped (optimizer.py:643 hands an int
to an assertion on ``.shape``); the fixture records that, the drop-in implements the evident intent and is checked against the
oracle in tests/test_shapes_gpu.py."""
import os

import numpy as np
import pytest
import torch

import golden_inputs as gi
from test_optimizer_gpu import _DS
from test_round2_gaps_gpu import _new_opt

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_optimized_variables_and_predict_after_three_cycles(smpl_struct, smpl_regs, tmp_path):
    post = np.load(os.path.join(ROOT, 'tests', 'golden', 'reference_post_cpu.npz'), allow_pickle=False)
    assert bytes(post['post_filtered_verts_raises']).decode() == 'AttributeError'
    fin = gi.fit_inputs()
    opt = _new_opt(smpl_struct, smpl_regs, tmp_path, fin)
    opt.init_optimized_variables(fin['pose2d'], fin['poses_smpl'], fin['betas_smpl'], fin['valid_smpl'], num_iter=0)
    e = opt.engine
    e.leaf('poses_T').copy_(torch.tensor(post['post_init_poses_T']).view(fin['T'], fin['N'], 3))
    e.leaf('zmax_lin').copy_(torch.tensor(post['post_init_zmax_lin']).view(-1))
    opt.scene_depth = fin['scene_depth']
    opt.update_scene_pointcloud(fin['scene_depth'], fin['scene_mask'])
    opt.fit(torch.utils.data.DataLoader(_DS(fin), batch_size=5, shuffle=False), num_iter=3)
    ov = opt.get_optimized_variables()
    for k in ['scale_factor', 'poses_T', 'poses_smpl', 'betas_smpl', 'valid_smpl', 'min_z', 'max_z']:
        want = post['post_ov_' + k]
        got = np.asarray(ov[k])
        assert got.shape == want.shape and got.dtype == want.dtype, (k, got.shape, want.shape, got.dtype, want.dtype)
        err = np.abs(got - want)
        assert (err > 1e-4).mean() <= 0.002 and err.max() <= 1e-3, (k, float((err > 1e-4).mean()), float(err.max()))
    # predict on the REFERENCE's variables: same inputs, so only the SMPL pass and the composition are compared
    verts, joints = opt.predict(post['post_ov_poses_T'][3], post['post_ov_poses_smpl'][3], post['post_ov_betas_smpl'][0],
                                post['post_ov_scale_factor'][0])
    verts, joints = np.asarray(verts), np.asarray(joints)
    assert verts.shape == (fin['N'], 6890, 3) and joints.shape == post['post_predict_joints'].shape
    np.testing.assert_allclose(verts[:, ::53], post['post_predict_verts_sub'], atol=2e-6)
    np.testing.assert_allclose(joints, post['post_predict_joints'], atol=2e-6)
    # the method the reference cannot run returns the filtered sequence here
    vf = opt.get_filtered_vertices_by_smpl()
    vf = vf.cpu().numpy() if isinstance(vf, torch.Tensor) else np.asarray(vf)
    assert vf.shape == (fin['T'], fin['N'], 6890, 3) and np.isfinite(vf).all()
