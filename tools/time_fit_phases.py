"""developer tool: wall time of opt.fit on C3 for several iteration counts (static cycles < 30, organic scene from 30 on)"""
import os, sys, tempfile, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tools')]
import bench
from mhhip import synthetic, synthetic_seq
from mhmocap.optimizer import SMPLDepthSequenceOptimizer
struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct)
K = synthetic.default_cam_K(bench.IMG, 60.0)
tmp = tempfile.mkdtemp()
opt0 = bench.build_optimizer(struct, regs, tmp, 200, 'cuda:0', K)
seq = synthetic_seq.make_sequence(opt0.SMPLPY.body_model, 4, 200, bench.IMG, 1003, cam_K=K)
c = bench.COEFS
for n in (30, 60, 250, 250):
    opt = SMPLDepthSequenceOptimizer(
        image_size=bench.IMG, num_frames=200, cam_K=K, device='cuda:0', smpl_model_parameters_path=tmp, smpl_data_struct=struct,
        proj2d_loss_coef=c['proj2d'], depth_loss_coef=c['depth'], silhouette_loss_coef=c['silhouette'],
        reg_velocity_coef=c['reg_velocity'], reg_verts_filter_coef=c['reg_verts_filter'], reg_poses_coef=c['reg_poses'],
        reg_scales_coef=c['reg_scales'], reg_contact_coef=c['reg_contact'], reg_foot_sliding_coef=c['reg_foot_sliding'])
    opt.init_optimized_variables(seq['pose2d'], seq['poses_smpl'], seq['betas_smpl'], seq['valid_smpl'], num_iter=100)
    dl = torch.utils.data.DataLoader(synthetic_seq.SequenceDataset(seq), batch_size=10, shuffle=False)
    opt._stage_from_dataloader(dl)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt.fit(dl, num_iter=n)
    torch.cuda.synchronize()
    print('fit(%d): %.1f ms' % (n, (time.perf_counter() - t0) * 1e3))
