"""BUILD-CONTAINER ONLY (needs /root/reference): wall time per cycle of the REFERENCE's own ``fit`` loop on the CPU
at the C3 shape (4 humans, 240x135, batch 10, nine terms), PyTorch3D replaced by the oracle rasteriser through the
stubs of tests/golden/make_golden_raster.py, on a 20-frame sample (the same sample size bench.py's cpu_baseline
uses).  Feeds BASELINE.md's "reference loop" CPU number; nothing here ships to the GPU box.

    python tools/time_reference_loop.py [frames] [cycles] [threads]
"""
import importlib
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'tests', 'golden'), ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd')):
    sys.path.insert(0, p)
sys.dont_write_bytecode = True
import golden_inputs as gi  # noqa: E402
import make_golden as mg  # noqa: E402
import make_golden_raster as mgr  # noqa: E402
from mhhip import synthetic  # noqa: E402
from oracle import lbs_oracle  # noqa: E402


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    threads = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    torch.set_num_threads(threads)
    sys.argv = ['x']
    struct = synthetic.make_smpl_struct(1)
    regs = synthetic.make_extra_regressors(1, struct)
    faces = np.asarray(struct.f).astype(np.int64)
    mgr._install_raster_stubs(faces)
    mg._ref_package()
    smpl = importlib.import_module('refmh.smpl')
    optim = importlib.import_module('refmh.optimizer')
    tmp = tempfile.mkdtemp()
    for k, fn in [('extra9', 'J_regressor_extra.npy'), ('h36m', 'J_regressor_h36m.npy'),
                  ('alphapose', 'SMPL_AlphaPose_Regressor_RMSprop_6.npy')]:
        np.save(os.path.join(tmp, fn), regs[k])
    optim.SMPL = lambda path, **kw: smpl.SMPL(None, data_struct=smpl.Struct(**struct.__dict__), **{
        **dict(J_reg_extra9_path=os.path.join(tmp, 'J_regressor_extra.npy'), J_reg_h36m17_path=os.path.join(tmp, 'J_regressor_h36m.npy'),
               J_reg_alphapose_path=os.path.join(tmp, 'SMPL_AlphaPose_Regressor_RMSprop_6.npy')), **kw})
    fin = mgr.make_inputs(lbs_oracle.BodyModel(struct, regs), T=frames, N=4, W=240, H=135, seed=61)
    # four bodies spread over the image (make_inputs places two; add two more tracks)
    c = gi.COEFS
    opt = optim.SMPLDepthSequenceOptimizer(
        image_size=(fin['W'], fin['H']), num_frames=frames, cam_K=fin['cam_K'], device='cpu', smpl_model_parameters_path=tmp,
        proj2d_loss_coef=c['proj2d'], depth_loss_coef=c['depth'], silhouette_loss_coef=c['silhouette'],
        reg_velocity_coef=c['reg_velocity'], reg_verts_filter_coef=c['reg_verts_filter'], reg_poses_coef=c['reg_poses'],
        reg_scales_coef=c['reg_scales'], reg_contact_coef=c['reg_contact'], reg_foot_sliding_coef=c['reg_foot_sliding'])

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return frames

        def __getitem__(self, i):
            return dict(images=fin['images'][i], depths=fin['depths'][i], seg_mask=fin['seg_mask'][i], backmasks=fin['backmasks'][i],
                        pose2d=fin['pose2d'][i], poses_smpl=fin['poses_smpl'][i], betas_smpl=fin['betas_smpl'][i],
                        valid_smpl=fin['valid_smpl'][i], idxs=i)
    t0 = time.perf_counter()
    opt.init_optimized_variables(fin['pose2d'], fin['poses_smpl'], fin['betas_smpl'], fin['valid_smpl'], num_iter=100)
    t_init = time.perf_counter() - t0
    opt.scene_depth = fin['scene_depth']
    opt.update_scene_pointcloud(fin['scene_depth'], fin['scene_mask'])
    dl = torch.utils.data.DataLoader(DS(), batch_size=10, shuffle=False)
    try:
        opt.fit(dl, num_iter=1)           # untimed first touch
    except UnboundLocalError:
        pass
    t0 = time.perf_counter()
    try:
        opt.fit(dl, num_iter=cycles)
    except UnboundLocalError:
        pass
    dt = (time.perf_counter() - t0) / cycles
    print('reference fit loop (oracle rasteriser in the pytorch3d stubs): %d frames x 4 humans, 240x135, %d threads: '
          '%.2f s per cycle -> %.1f s per cycle at 200 frames = %.5f it/s; warm-up (100 iterations) %.1f s'
          % (frames, threads, dt, dt * 200.0 / frames, frames / (dt * 200.0), t_init))


if __name__ == '__main__':
    main()
