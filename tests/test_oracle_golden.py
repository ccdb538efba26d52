"""Pins the oracle (oracle/*.py) to the reference: every comparison is against numbers the
reference's own CPU code produced in the build container (tests/golden/make_golden.py)."""
import numpy as np
import pytest
import torch

import golden_inputs as gi
from oracle import fit_oracle as fo
from oracle import lbs_oracle as lo


def close(a, b, atol, rtol=0.0):
    np.testing.assert_allclose(np.asarray(a), np.asarray(b), atol=atol, rtol=rtol)


def test_smpl_forward_all_outputs(golden, oracle_model):
    betas, poses = gi.lbs_inputs()
    out = lo.smpl_forward(oracle_model, torch.tensor(betas), torch.tensor(poses))
    close(out['verts'][:, ::53], golden['smpl_verts'], 2e-6)
    for k in ['j3d', 'joints_smpl24', 'joints_h36m17', 'joints_alphapose', 'joints_mupots']:
        close(out[k], golden['smpl_' + k], 2e-6)
    close(out['verts'].sum(1), golden['smpl_verts_sum'], 5e-3)
    close((out['verts'].double() ** 2).sum(1), golden['smpl_verts_sqsum'], 1e-2)
    close(out['joints_smpl24'], golden['smpl_chunked_joints'], 2e-6)


def test_smpl_autograd_gradients(golden, oracle_model):
    betas, poses = gi.lbs_inputs()
    tb = torch.tensor(betas, requires_grad=True)
    tp = torch.tensor(poses, requires_grad=True)
    out = lo.smpl_forward(oracle_model, tb, tp)
    rng = np.random.RandomState(3)
    wv = torch.tensor(rng.normal(0, 1, out['verts'].shape).astype(np.float32))
    wj = torch.tensor(rng.normal(0, 1, out['joints_alphapose'].shape).astype(np.float32))
    ((out['verts'] * wv).sum() + (out['joints_alphapose'] * wj).sum()).backward()
    g = golden['smpl_grad_betas']
    close(tb.grad, g, 2e-4 * np.abs(g).max())
    g = golden['smpl_grad_poses']
    # body 0 has theta == 0 exactly: d(rodrigues) there is dominated by the 1e-8 shift, compare loosely
    close(tp.grad[1:], g[1:], 2e-4 * np.abs(g[1:]).max())
    close(tp.grad[0, :66], g[0, :66], 2e-2 * np.abs(g[0]).max())
    assert np.all(tp.grad.numpy()[:, 66:] == 0) and np.all(g[:, 66:] == 0)   # hands never move (smpl.py:542-546)


def test_rodrigues(golden):
    close(lo.rodrigues(torch.tensor(gi.rodrigues_inputs())), golden['rodrigues'], 1e-6)


def test_camera(golden):
    pts, K, Kd = gi.projection_inputs()
    uv = fo.project_points(torch.tensor(pts), torch.tensor(K))
    close(uv, golden['proj_plain'], 1e-4)
    close(fo.project_points(torch.tensor(pts), torch.tensor(K), Kd), golden['proj_dist'], 1e-4)
    uvd = torch.cat([uv, torch.tensor(pts[..., 2:])], -1)
    close(fo.unproject_points(uvd, torch.tensor(K)), golden['unproj'], 1e-5)
    close(fo.calibration_matrix_ndc(1.0, 100.0, K[0], (240, 135)), golden['calib_land'], 1e-6)
    close(fo.calibration_matrix_ndc(1.0, 100.0, K[0], (135, 240)), golden['calib_port'], 1e-6)
    close(fo.calibration_matrix_ndc(1.0, 100.0, K[0], (256, 256)), golden['calib_sq'], 1e-6)
    close(fo.softplus(torch.tensor(np.linspace(-5, 9, 29).astype(np.float32))), golden['softplus'], 1e-6)


def test_image_losses(golden):
    pred, true, mask = gi.image_loss_inputs()
    tp = torch.tensor(pred, requires_grad=True)
    tt = torch.tensor(true, requires_grad=True)
    l = fo.avg_log_depth_loss(tp, tt, torch.tensor(mask))
    l.backward()
    close(l.detach(), golden['depth_loss'], 1e-6, 1e-5)
    close(tp.grad, golden['depth_loss_gpred'], 1e-7, 1e-4)
    close(tt.grad, golden['depth_loss_gtrue'], 1e-7, 1e-4)
    a = torch.tensor(pred[:, 0], requires_grad=True)
    l = fo.masked_mse_loss(a, torch.tensor(true[:, 0]), torch.tensor(mask[:, 0]))
    l.backward()
    close(l.detach(), golden['mse_loss'], 1e-6, 1e-5)
    close(a.grad, golden['mse_loss_grad'], 1e-8, 1e-5)


def test_erode_twice(golden):
    x = torch.tensor(gi.erode_inputs())
    np.testing.assert_array_equal(fo.erode3x3(fo.erode3x3(x)).numpy(), golden['erode2'])


def test_one_euro(golden):
    x = gi.one_euro_inputs()
    close(fo.one_euro_sequence(x, 0.01, 0.02), golden['one_euro_a'], 1e-6)
    close(fo.one_euro_sequence(x, 0.001, 0.5), golden['one_euro_b'], 1e-6)


LEAVES = ['poses_T', 'poses_smpl', 'betas_smpl', 'zmin_lin', 'zmax_lin', 'xscale_factor']


def _oracle_leaves(o):
    return dict(zip(LEAVES, [p.detach().numpy() for p in o.leaves()]))


def _new_oracle(oracle_model, fin, scene):
    stub = lambda v: (-torch.ones(v.shape[0], fin['H'], fin['W']) + 0.0 * v.sum(),
                      torch.zeros(v.shape[0], fin['H'], fin['W']) + 0.0 * v.sum())
    o = fo.SequenceOracle(oracle_model, (fin['W'], fin['H']), fin['T'], fin['cam_K'], coefs=gi.COEFS, rasteriser=stub)
    o.xscale = torch.zeros(1, fin['N'], 1, 1)
    return o


def _batches(fin, b=5):
    out = []
    for s in range(0, fin['T'], b):
        sl = slice(s, s + b)
        out.append(dict(idxs=torch.arange(s, min(s + b, fin['T'])), pose2d=torch.tensor(fin['pose2d'][sl]),
                        seg_mask=torch.tensor(fin['seg_mask'][sl]), depths=torch.tensor(fin['depths'][sl]),
                        poses_smpl=torch.tensor(fin['poses_smpl'][sl])))
    return out


def test_warmup_and_init(golden, oracle_model):
    fin = gi.fit_inputs()
    o = _new_oracle(oracle_model, fin, False)
    log = o.init_optimized_variables(fin['pose2d'], fin['poses_smpl'], fin['betas_smpl'], fin['valid_smpl'], num_iter=5)
    close(np.array(log, np.float32), golden['init_loss2d_log'], 0, 2e-5)
    got = _oracle_leaves(o)
    for n in LEAVES:
        close(got[n], golden['fit_init_' + n], 2e-5)


@pytest.mark.parametrize('scene', [False, True])
def test_fit_first_cycle_gradients(golden, oracle_model, scene):
    fin = gi.fit_inputs()
    o = _new_oracle(oracle_model, fin, scene)
    o.init_optimized_variables(fin['pose2d'], fin['poses_smpl'], fin['betas_smpl'], fin['valid_smpl'],
                               poses_T=golden['fit_init_poses_T'])
    if scene:
        o.update_scene_pointcloud(fin['scene_depth'], fin['scene_mask'])
        close(o.scene_pcd[0, 0], golden['scene_pcd'], 1e-5)
    o.cycle_grads(_batches(fin))
    pre = 'fitscene_k1_grad_' if scene else 'fit_k1_grad_'
    for n, p in zip(LEAVES, o.leaves()):
        if pre + n in golden.files:
            g = golden[pre + n]
            got = p.grad.numpy() if p.grad is not None else np.zeros_like(g)
            close(got, g, 3e-4 * max(np.abs(g).max(), 1e-6))


@pytest.mark.parametrize('k,scene', [(1, False), (5, False), (30, False), (5, True)])
def test_fit_k_cycles(golden, oracle_model, k, scene):
    fin = gi.fit_inputs()
    o = _new_oracle(oracle_model, fin, scene)
    o.init_optimized_variables(fin['pose2d'], fin['poses_smpl'], fin['betas_smpl'], fin['valid_smpl'],
                               poses_T=golden['fit_init_poses_T'])
    if scene:
        o.update_scene_pointcloud(fin['scene_depth'], fin['scene_mask'])
    o.fit(_batches(fin), k)
    got = _oracle_leaves(o)
    pre = 'fitscene_k%d_' % k if scene else 'fit_k%d_' % k
    tol = {1: 2e-5, 5: 2e-4, 30: 2e-3}[k]
    for n in LEAVES:
        close(got[n], golden[pre + n], tol)
    if k == 1 and not scene:
        ov = o.optimized_variables()
        close(ov['min_z'], golden['optvar_min_z'], 1e-5)
        close(ov['max_z'], golden['optvar_max_z'], 1e-4)
        close(ov['scale_factor'], golden['optvar_scale'], 1e-6)


def test_fit_logs_before_chaos(golden, oracle_model):
    """Per-cycle loss logs of the scene run: identical trajectories for the first 30 cycles."""
    fin = gi.fit_inputs()
    o = _new_oracle(oracle_model, fin, True)
    o.init_optimized_variables(fin['pose2d'], fin['poses_smpl'], fin['betas_smpl'], fin['valid_smpl'],
                               poses_T=golden['fit_init_poses_T'])
    o.update_scene_pointcloud(fin['scene_depth'], fin['scene_mask'])
    log = o.fit(_batches(fin), 30)
    for key in ['loss_pose24j', 'loss_silhouette', 'reg_ref_poses', 'reg_scale', 'reg_vel', 'reg_contact',
                'reg_foot_sliding']:
        ref = golden['fitlong_log_' + key][:30]
        mine = np.array([l[key] for l in log], np.float32)
        close(mine, ref, 1e-6, 5e-3)
    assert golden['fitlong_log_reg_foot_sliding'][:30].max() > 0      # the gate is exercised


def test_cycle50_filters_and_gradients(golden, oracle_model):
    """State entering cycle 50 (after 50 reference steps) -> one-euro filters, filtered-vertex
    term and the full gradient of cycle 50 (optimizer.py:383-392, 560-575)."""
    fin = gi.fit_inputs()
    o = _new_oracle(oracle_model, fin, True)
    o.init_optimized_variables(fin['pose2d'], fin['poses_smpl'], fin['betas_smpl'], fin['valid_smpl'],
                               poses_T=golden['fit_init_poses_T'])
    o.update_scene_pointcloud(fin['scene_depth'], fin['scene_mask'])
    with torch.no_grad():
        for n, p in zip(LEAVES, o.leaves()):
            p.copy_(torch.tensor(golden['fitlong_k50_' + n]))
    o.update_filters(cpu_alias_quirk=True)     # fixtures come from the reference's CPU device
    close(o.pT_filt, golden['fitlong_pT_filtered'], 2e-6)
    close(o.v_filt[:, :, ::53], golden['fitlong_verts_filtered_sub'], 5e-6)
    log = o.cycle_grads(_batches(fin))
    close(log['reg_filter_verts'], golden['fitlong_c50_reg_filter_verts'], 0, 1e-3)
    close(log['reg_foot_sliding'], golden['fitlong_c50_reg_foot_sliding'], 1e-7, 1e-3)
    for n, p in zip(LEAVES, o.leaves()):
        g = golden['fitlong_c50_grad_' + n]
        close(p.grad.numpy(), g, 3e-4 * max(np.abs(g).max(), 1e-6))
