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
