"""Golden fixture for the evaluator the optimiser's output goes to (eval_mupots.py:18-31): the reference's
``compute_smpl_pred_error_3dproj`` (evaluate.py:180-296) with the reference's own CPU ``SMPL`` as ``SMPLPY``, plus
``masked_average_error`` / ``masked_average_pck`` of its outputs, on three small synthetic cases:

  mupots   17-joint ground truth (float64), K = 3 reference persons against N = 2 predictions (rows compacted to the
           matched pairs), one scale per person, persons swapped, a person with no visible joint, hidden roots
  dist     K = 2 against N = 3, lens distortion, one scale per (frame, person), float32 ground truth
  panoptic 19-joint CMU-Panoptic ground truth (the AlphaPose regressor route, both layout maps)

Inputs (the ``get_optimized_variables`` dict, ground truth, visibility) are stored next to the outputs, and so are the
sparse joints of the reference's body model (for the host-logic test, which runs without a device).
Only in the build container (``/root/reference``); writes numbers only.

    python tests/golden/make_golden_eval.py
"""
import importlib
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), 'scene-aware-3d-multi-human_amd'))
import make_golden as mg  # noqa: E402
from mhhip import synthetic  # noqa: E402

CASES = {'mupots': dict(T=12, N=2, K=3, J=17, kd=None, per_frame_scale=False, gt_dtype=np.float64, seed=11),
         'dist': dict(T=9, N=3, K=2, J=17, kd=[0.08, -0.02, 0.004, -0.003, 0.001], per_frame_scale=True, gt_dtype=np.float32, seed=12),
         'panoptic': dict(T=7, N=2, K=2, J=19, kd=None, per_frame_scale=False, gt_dtype=np.float32, seed=13)}
CAM_K = np.array([[310.0, 0, 120.0], [0, 305.0, 67.5], [0, 0, 1]], np.float32)


def main():
    assert os.path.isdir(mg.REF), 'reference not present: fixtures can only be regenerated in the build container'
    sys.argv = ['x']
    mg._install_stubs()
    mg._ref_package()
    smpl = importlib.import_module('refmh.smpl')
    ev = importlib.import_module('refmh.evaluate')
    torch.set_num_threads(8)
    struct = synthetic.make_smpl_struct(1)
    regs = synthetic.make_extra_regressors(1, struct)
    import tempfile
    tmp = tempfile.mkdtemp()
    p = {}
    for k, fn in [('extra9', 'J_regressor_extra.npy'), ('h36m', 'J_regressor_h36m.npy'),
                  ('alphapose', 'SMPL_AlphaPose_Regressor_RMSprop_6.npy'), ('mupots', 'SMPL_MuPoTs_Regressor.npy')]:
        p[k] = os.path.join(tmp, fn)
        np.save(p[k], regs[k])
    body = smpl.SMPL(None, J_reg_extra9_path=p['extra9'], J_reg_h36m17_path=p['h36m'], J_reg_alphapose_path=p['alphapose'],
                     J_reg_mupots_path=p['mupots'], data_struct=smpl.Struct(**struct.__dict__))
    out = {'cam_K': CAM_K}
    for tag, c in CASES.items():
        rng = np.random.RandomState(c['seed'])
        T, N, K, J = c['T'], c['N'], c['K'], c['J']
        ov = {'poses_smpl': (0.25 * rng.randn(T, N, 72)).astype(np.float32),
              'betas_smpl': (0.5 * rng.randn(T, N, 10)).astype(np.float32),
              'poses_T': np.concatenate([rng.uniform(-1.5, 1.5, (T, N, 1, 1)), rng.uniform(0.7, 1.1, (T, N, 1, 1)),
                                         rng.uniform(3, 8, (T, N, 1, 1))], -1).astype(np.float32),
              'scale_factor': (1.1 ** rng.uniform(-1, 1, (T if c['per_frame_scale'] else 1, N, 1, 1))).astype(np.float32),
              'valid_smpl': np.ones((T, N, 1), np.float32)}
        res = body(betas=ov['betas_smpl'].reshape(-1, 10), poses=ov['poses_smpl'].reshape(-1, 72))
        sparse = {k: res[k].cpu().numpy() for k in ['joints_mupots', 'joints_alphapose']}
        sc = np.tile(ov['scale_factor'], (T, 1, 1, 1)) if ov['scale_factor'].shape[0] == 1 else ov['scale_factor']
        if J == 17:
            base = sc * sparse['joints_mupots'].reshape(T, N, 17, 3) + ov['poses_T']
        else:
            # a 19-joint ground truth whose 15 mapped joints sit near the prediction's: invert the permutation of the map
            j15 = ev.map_alphapose_to_mupots15j(sparse['joints_alphapose'].reshape(T * N, 17, 3)).reshape(T, N, 15, 3)
            j15 = sc * j15 + ov['poses_T']
            base = np.zeros((T, N, 19, 3), np.float32)
            for m, (_, src) in enumerate(ev.cmu_panoptic_to_mupots15j_map):
                base[:, :, src[0]] = j15[:, :, m]
            base[:, :, 15:] = j15[:, :, :4] + 0.1
        # reference persons: the predictions in reversed order + 3 cm of noise; further persons stand elsewhere
        gt = np.zeros((T, K, J, 3))
        for k in range(K):
            if k < N:
                gt[:, k] = base[:, N - 1 - k] + 0.03 * rng.randn(T, J, 3)
            else:
                gt[:, k] = base[:, 0] + np.array([2.5, 0.1, 1.5]) + 0.03 * rng.randn(T, J, 3)
        vis = rng.uniform(0, 1, (T, K, J, 1))
        vis[vis > 0.35] = 1.0                                    # most joints visible, the rest in (0, 0.35]
        vis[2, 0] = 0.0                                           # a reference person without any visible joint
        vis[3, :, 14 if J == 17 else 2] = 0.0                     # hidden roots (MuPoTs joint 14 = Panoptic joint 2)
        if K > N:
            vis[5, 1] = 0.0
        gt, vis = gt.astype(c['gt_dtype']), vis.astype(c['gt_dtype'])
        m = ev.compute_smpl_pred_error_3dproj({k: v.copy() for k, v in ov.items()}, gt.copy(), vis.copy(), body, CAM_K.copy(),
                                              Kd=None if c['kd'] is None else np.array(c['kd'], np.float32))
        for k, v in ov.items():
            out['%s_ov_%s' % (tag, k)] = v
        out[tag + '_ref_poses3d'], out[tag + '_visibility'] = gt, vis
        out[tag + '_kd'] = np.zeros(0, np.float32) if c['kd'] is None else np.array(c['kd'], np.float32)
        for k, v in sparse.items():
            out['%s_%s' % (tag, k)] = v
        for k, v in m.items():
            assert v.dtype == np.float32
            out['%s_out_%s' % (tag, k)] = v
        out[tag + '_summary'] = np.array([
            ev.masked_average_error(m['abs_dist'], m['valid_joints']), ev.masked_average_error(m['rel_dist'], m['valid_joints']),
            ev.masked_average_error(m['abs_root_pos_err'], m['valid_root']), ev.masked_average_pck(m['rel_dist'], m['valid_joints'], 0.15),
            ev.masked_average_pck(m['abs_root_pos_err'], m['valid_root'], 0.25),
            ev.masked_average_error(m['abs_jitter'], m['valid_joints'])], np.float64)
        print(tag, 'mm abs %.2f rel %.2f mrpe %.2f pck %.3f ap25 %.3f jitter %.2f' % tuple(
            out[tag + '_summary'] * [1000, 1000, 1000, 1, 1, 1000]))
    dst = os.path.join(HERE, 'reference_eval_cpu.npz')
    np.savez_compressed(dst, **out)
    print('wrote', dst, os.path.getsize(dst), 'bytes')


if __name__ == '__main__':
    main()
