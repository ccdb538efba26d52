"""Odd shapes through the whole drop-in optimiser (no parity claim: finiteness, shapes, contracts): one human, a ragged
last batch, portrait / square images, a single frame, 35 cycles so that the device scene path and the filters engage."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('T,N,W,H,batch', [(3, 1, 50, 70, 2), (5, 3, 64, 64, 4), (1, 2, 80, 45, 1), (7, 2, 90, 60, 3)])
def test_fit_runs_on_odd_shapes(smpl_struct, smpl_regs, tmp_path, T, N, W, H, batch):
    from mhhip import synthetic, synthetic_seq
    from mhmocap.optimizer import SMPLDepthSequenceOptimizer
    for k, fn in [('extra9', 'J_regressor_extra.npy'), ('h36m', 'J_regressor_h36m.npy'),
                  ('alphapose', 'SMPL_AlphaPose_Regressor_RMSprop_6.npy')]:
        np.save(str(tmp_path / fn), smpl_regs[k])
    K = synthetic.default_cam_K((W, H), 60.0)
    opt = SMPLDepthSequenceOptimizer(image_size=(W, H), num_frames=T, fov=60, device='cuda:0',
                                     smpl_model_parameters_path=str(tmp_path), smpl_data_struct=smpl_struct, cam_K=K)
    seq = synthetic_seq.make_sequence(opt.SMPLPY.body_model, N, T, (W, H), 5 + T, cam_K=K, z_range=(2.6, 3.6))
    opt.init_optimized_variables(seq['pose2d'], seq['poses_smpl'], seq['betas_smpl'], seq['valid_smpl'], num_iter=10)
    dl = torch.utils.data.DataLoader(synthetic_seq.SequenceDataset(seq), batch_size=batch, shuffle=False)
    log = opt.fit(dl, num_iter=35)
    assert len(log) == 35
    for row in log:
        for k, v in row.items():
            assert np.isfinite(float(v)), (k, v)
    out = opt.get_optimized_variables()
    assert out['poses_T'].shape == (T, N, 1, 3) and out['poses_smpl'].shape == (T, N, 72)
    assert out['betas_smpl'].shape == (1, N, 10) and out['scale_factor'].shape == (1, N, 1, 1)
    assert out['scene_depth'].shape == (H, W) and np.isfinite(out['scene_depth']).all()
    assert out['scene_img'].shape == (H, W, 3)
    for k in ('poses_T', 'poses_smpl', 'betas_smpl', 'min_z', 'max_z'):
        assert np.isfinite(out[k]).all(), k


def test_filtered_vertices_by_smpl_and_predict(smpl_struct, smpl_regs, oracle_model, tmp_path):
    """a23 / a7: get_filtered_vertices_by_smpl (optimizer.py:639-661) and predict (:133-143) against the CPU oracle"""
    from mhhip import synthetic, synthetic_seq
    from mhmocap.optimizer import SMPLDepthSequenceOptimizer
    from mhmocap.one_euro_filter import OneEuroFilter
    from oracle import lbs_oracle as lo
    for k, fn in [('extra9', 'J_regressor_extra.npy'), ('h36m', 'J_regressor_h36m.npy'),
                  ('alphapose', 'SMPL_AlphaPose_Regressor_RMSprop_6.npy')]:
        np.save(str(tmp_path / fn), smpl_regs[k])
    T, N, W, H = 6, 2, 64, 48
    K = synthetic.default_cam_K((W, H), 60.0)
    opt = SMPLDepthSequenceOptimizer(image_size=(W, H), num_frames=T, fov=60, device='cuda:0',
                                     smpl_model_parameters_path=str(tmp_path), smpl_data_struct=smpl_struct, cam_K=K)
    seq = synthetic_seq.make_sequence(opt.SMPLPY.body_model, N, T, (W, H), 12, cam_K=K, z_range=(2.6, 3.6))
    opt.init_optimized_variables(seq['pose2d'], seq['poses_smpl'], seq['betas_smpl'], seq['valid_smpl'], num_iter=3)
    got = opt.get_filtered_vertices_by_smpl()
    got = got.cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    assert got.shape == (T, N, 6890, 3)
    out = opt.get_optimized_variables()
    pT, pose = out['poses_T'].copy(), out['poses_smpl'].copy()
    fT = OneEuroFilter(0, pT[0], dx0=0 * pT[0], min_cutoff=0.004, beta=0.7, d_cutoff=1.0)
    fP = OneEuroFilter(0, pose[0], dx0=0 * pose[0], min_cutoff=0.1, beta=0.1, d_cutoff=1.0)
    for i in range(1, T):
        pT[i] = fT(i / 25, pT[i])
        pose[i] = fP(i / 25, pose[i])
    be = torch.tensor(out['betas_smpl']).expand(T, N, 10).reshape(-1, 10)
    ref = lo.smpl_forward(oracle_model, be, torch.tensor(pose).view(-1, 72))
    want = out['scale_factor'].reshape(1, N, 1, 1) * ref['verts'].view(T, N, -1, 3).numpy() + pT
    np.testing.assert_allclose(got, want, atol=2e-5)
    verts, joints = opt.predict(out['poses_T'][0], out['poses_smpl'][0], np.tile(out['betas_smpl'][0], (1, 1)), out['scale_factor'][0])
    ref0 = lo.smpl_forward(oracle_model, torch.tensor(out['betas_smpl'][0]), torch.tensor(out['poses_smpl'][0]))
    np.testing.assert_allclose(verts, out['scale_factor'][0] * ref0['verts'].numpy() + out['poses_T'][0], atol=2e-5)
    np.testing.assert_allclose(joints, out['scale_factor'][0] * ref0['joints_alphapose'].numpy() + out['poses_T'][0], atol=2e-5)
