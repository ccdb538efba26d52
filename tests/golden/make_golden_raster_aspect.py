"""Golden fixture for the rasterised terms at OTHER ASPECT RATIOS: a portrait image (54 x 90, two humans) and a square one
(64 x 64, three humans).  The reference converts its intrinsics to PyTorch3D's NDC convention with one branch per aspect
(transforms.py:222-255: the longer side spans more than [-1, 1]); reference_raster_cpu.npz only exercises the landscape branch.

Same construction as make_golden_raster.py: the reference's own ``fit`` runs around stub ``MeshRasterizer`` / ``MeshRenderer``
classes that take the NDC coordinates from the reference's own camera objects and the face selection from the oracle; recorded
per variant: the rendered inputs, the leaves after the warm-up (20 iterations), per-leaf gradients and leaves after cycle 1, the
depth loss of every batch and every silhouette-loss call.  Only in the build container (``/root/reference``); numbers only.

    python tests/golden/make_golden_raster_aspect.py
"""
import importlib
import os
import sys
import tempfile

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_raster as mgr  # noqa: E402
import make_golden as mg  # noqa: E402
import golden_inputs as gi  # noqa: E402
from mhhip import synthetic  # noqa: E402
from oracle import lbs_oracle  # noqa: E402

VARIANTS = {'por': dict(T=20, N=2, W=54, H=90, seed=43), 'sq': dict(T=20, N=3, W=64, H=64, seed=44)}


def main():
    assert os.path.isdir(mg.REF), 'reference not present: fixtures can only be regenerated in the build container'
    sys.argv = ['x']
    struct = synthetic.make_smpl_struct(1)
    regs = synthetic.make_extra_regressors(1, struct)
    faces = np.asarray(struct.f).astype(np.int64)
    mgr._install_raster_stubs(faces)
    mg._ref_package()
    smpl = importlib.import_module('refmh.smpl')
    losses = importlib.import_module('refmh.losses')
    optim = importlib.import_module('refmh.optimizer')
    torch.manual_seed(0)
    torch.set_num_threads(8)
    tmp = tempfile.mkdtemp()
    paths = {}
    for k, fn in [('extra9', 'J_regressor_extra.npy'), ('h36m', 'J_regressor_h36m.npy'),
                  ('alphapose', 'SMPL_AlphaPose_Regressor_RMSprop_6.npy')]:
        paths[k] = os.path.join(tmp, fn)
        np.save(paths[k], regs[k])

    def build_smpl(model_path=None, **kw):
        kw.setdefault('J_reg_extra9_path', paths['extra9'])
        kw.setdefault('J_reg_h36m17_path', paths['h36m'])
        kw.setdefault('J_reg_alphapose_path', paths['alphapose'])
        return smpl.SMPL(model_path, data_struct=smpl.Struct(**struct.__dict__), **kw)

    optim.SMPL = lambda path, **kw: build_smpl(**kw)
    rec = {'depth': [], 'sil': []}
    real_depth, real_mse = losses.build_avg_depth_loss_fn, losses.build_masked_mse_loss_fn

    def rec_depth(*a, **kw):
        fn = real_depth(*a, **kw)

        def wrapped(pred, true, mask):
            v = fn(pred, true, mask)
            rec['depth'].append(float(v.detach()))
            return v
        return wrapped

    def rec_mse(*a, **kw):
        fn = real_mse(*a, **kw)

        def wrapped(a_, b_, m_):
            v = fn(a_, b_, m_)
            rec['sil'].append(float(v.detach()))
            return v
        return wrapped

    optim.build_avg_depth_loss_fn = rec_depth
    optim.build_masked_mse_loss_fn = rec_mse
    c = gi.COEFS
    coef_kw = dict(proj2d_loss_coef=c['proj2d'], depth_loss_coef=c['depth'], silhouette_loss_coef=c['silhouette'],
                   reg_velocity_coef=c['reg_velocity'], reg_verts_filter_coef=c['reg_verts_filter'],
                   reg_poses_coef=c['reg_poses'], reg_scales_coef=c['reg_scales'], reg_contact_coef=c['reg_contact'],
                   reg_foot_sliding_coef=c['reg_foot_sliding'])
    names = ['poses_T', 'poses_smpl', 'betas_smpl', 'zmin_lin', 'zmax_lin', 'xscale_factor']
    out = {}
    for tag, dims in VARIANTS.items():
        fin = mgr.make_inputs(lbs_oracle.BodyModel(struct, regs), **dims)
        fin['images'][:] = 0                          # not read by one cycle; random bytes would be most of the file
        for k in ['cam_K', 'pose2d', 'depths', 'poses_smpl', 'betas_smpl', 'valid_smpl', 'trans_gt', 'scene_depth']:
            out['%s_in_%s' % (tag, k)] = np.asarray(fin[k], np.float32)
        out[tag + '_in_seg_mask'] = fin['seg_mask'].astype(np.uint8)
        out[tag + '_in_images'] = fin['images']
        out[tag + '_in_backmasks'] = fin['backmasks'].astype(np.uint8)
        out[tag + '_in_scene_mask'] = fin['scene_mask'].astype(np.uint8)
        out[tag + '_in_dims'] = np.array([fin['T'], fin['N'], fin['H'], fin['W']], np.int64)

        class DS(torch.utils.data.Dataset):
            def __len__(self):
                return fin['T']

            def __getitem__(self, i):
                return dict(images=fin['images'][i], depths=fin['depths'][i], seg_mask=fin['seg_mask'][i],
                            backmasks=fin['backmasks'][i], pose2d=fin['pose2d'][i], poses_smpl=fin['poses_smpl'][i],
                            betas_smpl=fin['betas_smpl'][i], valid_smpl=fin['valid_smpl'][i], idxs=i)

        rec['depth'], rec['sil'] = [], []
        del mgr.CALLS['raster'][:], mgr.CALLS['render'][:]
        opt = optim.SMPLDepthSequenceOptimizer(image_size=(fin['W'], fin['H']), num_frames=fin['T'], cam_K=fin['cam_K'],
                                               device='cpu', smpl_model_parameters_path=tmp, **coef_kw)
        opt.init_optimized_variables(fin['pose2d'], fin['poses_smpl'], fin['betas_smpl'], fin['valid_smpl'], num_iter=20)
        for n in names:
            out['%s_init_%s' % (tag, n)] = getattr(opt, n).detach().numpy().copy()
        opt.scene_depth = fin['scene_depth']
        opt.update_scene_pointcloud(fin['scene_depth'], fin['scene_mask'])
        try:
            opt.fit(torch.utils.data.DataLoader(DS(), batch_size=5, shuffle=False), num_iter=1)
        except UnboundLocalError:
            pass                                       # optimizer.py:595 quirk for num_iter <= 30
        for n in names:
            out['%s_k1_%s' % (tag, n)] = getattr(opt, n).detach().numpy().copy()
            if getattr(opt, n).grad is not None:
                out['%s_k1_grad_%s' % (tag, n)] = getattr(opt, n).grad.numpy().copy()
        out[tag + '_k1_loss_depth_per_batch'] = np.array(rec['depth'], np.float32)
        out[tag + '_k1_loss_sil_calls'] = np.array(rec['sil'], np.float32)
        sizes = set(cl[0] for cl in mgr.CALLS['raster'] + mgr.CALLS['render'])
        assert sizes == {(fin['H'], fin['W'])}, sizes              # the reference hands (H, W) to RasterizationSettings
        assert (np.array(rec['depth']) > 0).all() and len(rec['sil']) > 0, 'bodies must cover supervised pixels'
        print(tag, dims, 'loss_depth per batch', rec['depth'], 'sil calls', len(rec['sil']), 'sum', float(np.sum(rec['sil'])))
    path = os.path.join(HERE, 'reference_raster_aspect_cpu.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB,', len(out), 'arrays')


if __name__ == '__main__':
    main()
