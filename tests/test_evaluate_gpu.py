"""The reference's evaluation step on the drop-in: ``mhmocap.evaluate.compute_smpl_pred_error_3dproj`` with the HIP body
model as ``SMPLPY`` (what ``eval_mupots.py:20-22`` does with ``dataset.SMPLPY``) against the reference's own evaluator run
with its own CPU body model (tests/golden/reference_eval_cpu.npz).  Same matches, every distance within the LBS
tolerance (1e-5 m on the sparse joints), the six reported numbers within 1e-5 relative."""
import numpy as np
import pytest

from test_evaluate_host import CASES, OUT, case_inputs, golden_eval  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def body(smpl_struct, smpl_regs, tmp_path_factory):
    from mhmocap.smpl import SMPL
    d = tmp_path_factory.mktemp('regs')
    p = {}
    for k, fn in [('extra9', 'J_regressor_extra.npy'), ('h36m', 'J_regressor_h36m.npy'),
                  ('alphapose', 'SMPL_AlphaPose_Regressor_RMSprop_6.npy'), ('mupots', 'SMPL_MuPoTs_Regressor.npy')]:
        p[k] = str(d / fn)
        np.save(p[k], smpl_regs[k])
    return SMPL(None, J_reg_extra9_path=p['extra9'], J_reg_h36m17_path=p['h36m'], J_reg_alphapose_path=p['alphapose'],
                J_reg_mupots_path=p['mupots'], data_struct=smpl_struct).to('cuda:0')


@pytest.mark.parametrize('tag', CASES)
def test_evaluator_on_the_device_body_model(golden_eval, body, tag):
    from mhmocap import evaluate as ev
    g = golden_eval
    ov, gt, vis, kd = case_inputs(g, tag)
    res = body(betas=ov['betas_smpl'].reshape(-1, 10), poses=ov['poses_smpl'].reshape(-1, 72))
    for k in ['joints_mupots', 'joints_alphapose']:
        assert res[k].is_cuda
        np.testing.assert_allclose(res[k].cpu().numpy(), g['%s_%s' % (tag, k)], atol=1e-5, rtol=0)
    m = ev.compute_smpl_pred_error_3dproj(ov, gt, vis, body, g['cam_K'], Kd=kd)
    for k in OUT:
        want = g['%s_out_%s' % (tag, k)]
        if k.startswith('valid'):
            assert np.array_equal(m[k], want), k
        else:
            np.testing.assert_allclose(m[k], want, rtol=0, atol=4e-5, err_msg=k)       # differences of two 1e-5 joints, scaled
    s = [ev.masked_average_error(m['abs_dist'], m['valid_joints']), ev.masked_average_error(m['rel_dist'], m['valid_joints']),
         ev.masked_average_error(m['abs_root_pos_err'], m['valid_root']), ev.masked_average_pck(m['rel_dist'], m['valid_joints'], 0.15),
         ev.masked_average_pck(m['abs_root_pos_err'], m['valid_root'], 0.25), ev.masked_average_error(m['abs_jitter'], m['valid_joints'])]
    np.testing.assert_allclose(s, g[tag + '_summary'], rtol=1e-5)


def test_evaluator_reads_what_fit_leaves_behind(smpl_struct, smpl_regs, tmp_path, body):
    """predict.py:344-347 -> eval_mupots.py:18-31 end to end on the drop-in: ``get_optimized_variables()`` of an optimiser
    after ``fit`` goes straight into the evaluator; against a ground truth made of its own joints the errors are zero and
    every prediction is matched to itself, whatever the order the ground truth lists the persons in."""
    import torch
    import golden_inputs as gi
    from mhmocap import evaluate as ev
    from mhmocap.optimizer import SMPLDepthSequenceOptimizer
    fin = gi.fit_inputs()
    for k, fn in [('extra9', 'J_regressor_extra.npy'), ('h36m', 'J_regressor_h36m.npy'),
                  ('alphapose', 'SMPL_AlphaPose_Regressor_RMSprop_6.npy')]:
        np.save(str(tmp_path / fn), smpl_regs[k])
    opt = SMPLDepthSequenceOptimizer(image_size=(fin['W'], fin['H']), num_frames=fin['T'], cam_K=fin['cam_K'], device='cuda:0',
                                     smpl_model_parameters_path=str(tmp_path), smpl_data_struct=smpl_struct, use_rasteriser=True)
    opt.init_optimized_variables(fin['pose2d'], fin['poses_smpl'], fin['betas_smpl'], fin['valid_smpl'], num_iter=5)

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return fin['T']

        def __getitem__(self, i):
            return dict(images=fin['images'][i], depths=fin['depths'][i], seg_mask=fin['seg_mask'][i], backmasks=fin['backmasks'][i],
                        pose2d=fin['pose2d'][i], poses_smpl=fin['poses_smpl'][i], betas_smpl=fin['betas_smpl'][i],
                        valid_smpl=fin['valid_smpl'][i], idxs=i)

    opt.fit(torch.utils.data.DataLoader(DS(), batch_size=5, shuffle=False), num_iter=3)
    ov = opt.get_optimized_variables()
    T, N = ov['poses_T'].shape[:2]
    assert ov['betas_smpl'].shape == (1, N, 10)
    ov['betas_smpl'] = np.repeat(ov['betas_smpl'], T, axis=0)                  # eval_mupots.py:118-119
    j = body(betas=ov['betas_smpl'].reshape(-1, 10), poses=ov['poses_smpl'].reshape(-1, 72))['joints_mupots'].cpu().numpy()
    gt = (ov['scale_factor'] * j.reshape(T, N, 17, 3) + ov['poses_T'])[:, ::-1].copy()             # persons listed in reverse
    m = ev.compute_smpl_pred_error_3dproj(ov, gt, np.ones((T, N, 17, 1), np.float32), body, np.asarray(fin['cam_K'], np.float32))
    assert m['valid_joints'].all() and m['valid_root'].all()
    for k in ['abs_dist', 'rel_dist', 'abs_root_pos_err', 'abs_jitter']:
        assert m[k].max() < 1e-5, (k, m[k].max())
