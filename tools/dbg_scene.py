import sys, os, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
from mhhip import synthetic, synthetic_seq
from mhmocap.optimizer import SMPLDepthSequenceOptimizer
from oracle import scene_oracle as scene_host
import golden_inputs as gi
struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct)
tmp = tempfile.mkdtemp()
for k, fn in [('extra9', 'J_regressor_extra.npy'), ('h36m', 'J_regressor_h36m.npy'), ('alphapose', 'SMPL_AlphaPose_Regressor_RMSprop_6.npy')]:
    np.save(os.path.join(tmp, fn), regs[k])
T, N, W, H, batch = 8, 2, 120, 68, 4
K = synthetic.default_cam_K((W, H), 60.0)
opt = SMPLDepthSequenceOptimizer(image_size=(W, H), num_frames=T, fov=60, device='cuda:0', smpl_model_parameters_path=tmp, smpl_data_struct=struct, scene_update='none', cam_K=K)
seq = synthetic_seq.make_sequence(opt.SMPLPY.body_model, N, T, (W, H), 77, cam_K=K, z_range=(2.6, 3.6))
opt.init_optimized_variables(seq['pose2d'], seq['poses_smpl'], seq['betas_smpl'], seq['valid_smpl'], num_iter=5)
dl = torch.utils.data.DataLoader(synthetic_seq.SequenceDataset(seq), batch_size=batch, shuffle=False)
opt._stage_from_dataloader(dl)
e = opt.engine
depths = scene_host.target_depths(e)
_, ma_depth, ma_mask = scene_host.aggregate_scene_median(depths, None, opt._backmasks)
want = scene_host.postprocess_depthmap(ma_depth, ma_mask, use_bilateral_filter=True)
e.scene_device_setup(opt._backmasks)
e.scene_device_update()
got, gmask, pts = e.scene_device_result()
d = e._scene_dev
md = d['ma_depth'].cpu().numpy()
print('backmask dtype', opt._backmasks.dtype, opt._backmasks.shape, 'valid frac', (opt._backmasks != 0).mean())
print('mask equal', (gmask == ma_mask).all(), 'median max rel diff', np.abs(md - ma_depth)[ma_mask].max() / ma_depth[ma_mask].max(), 'host data at masked', ma_depth[~ma_mask][:5])
print('final diff: frac > 1e-4', (np.abs(got - want) > 1e-4 * np.maximum(1, np.abs(want))).mean(), 'max', np.abs(got - want).max())
print(got[30, 50:56], want[30, 50:56], ma_depth[30, 50:56])
