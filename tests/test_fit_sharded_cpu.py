"""world_size-2 and -3 (gloo, CPU; 3 ranks: uneven blocks 6/3/3, a middle rank) run of the DROP-IN's ``fit`` (mhmocap/optimizer.py) against the single-process run: the
orchestration the 8-GPU run of ``predict_mupots.py`` under torchrun goes through -- every rank is handed the same
whole-sequence inputs, keeps its contiguous block of frames (block boundaries at batch multiples), the replicated
shape / scale leaves are broadcast from rank 0, each cycle exchanges the halos and all-reduces the shared gradient
tail (mhhip/sharded.py), the one-euro filter state is handed rank 0 -> 1, ``get_optimized_variables`` gathers the
whole sequence on every rank.  The compute engine is the torch-CPU stand-in of tests/cpu_shard_engine.py (raster-free
terms), plugged in by subclassing the optimiser (tests/cpu_shard_engine.py::cpu_optimizer_class); the HIP engine runs the same driver
(tests/test_sharded_gpu.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from netutil import free_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T, N, W, H, BATCH, CYCLES = 12, 2, 48, 32, 3, 33
COEFS = dict(proj2d_loss_coef=1.0, reg_poses_coef=0.002, reg_scales_coef=1e-4, reg_velocity_coef=0.05,
             reg_verts_filter_coef=0.002, depth_loss_coef=0.0, silhouette_loss_coef=0.0, reg_contact_coef=0.0,
             reg_foot_sliding_coef=0.0)


def _paths():
    for p in (ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
        if p not in sys.path:
            sys.path.insert(0, p)


def _inputs():
    _paths()
    from mhhip import synthetic
    st = synthetic.make_smpl_struct(1)
    regs = synthetic.make_extra_regressors(1, st)
    sp = synthetic.make_sequence_params(N, T, 78)
    rng = np.random.RandomState(6)
    K = synthetic.default_cam_K((W, H), 60.0)
    pose2d = np.zeros((T, N, 17, 3), np.float32)
    pose2d[..., 0] = rng.uniform(0, W, (T, N, 17)); pose2d[..., 1] = rng.uniform(0, H, (T, N, 17))
    pose2d[..., 2] = rng.uniform(0.3, 1.0, (T, N, 17))
    return st, regs, sp, K, pose2d


class _DS(torch.utils.data.Dataset):
    def __init__(self, sp, pose2d):
        self.sp, self.p = sp, pose2d

    def __len__(self):
        return T

    def __getitem__(self, i):
        return dict(pose2d=self.p[i], poses_smpl=self.sp['poses_init'][i], betas_smpl=self.sp['betas_init'][i],
                    valid_smpl=self.sp['valid'][i], idxs=i)


def _run_fit(tmp):
    """the caller's view: exactly what predict.py does with the optimiser (predict.py:290-306, 332-347)"""
    st, regs, sp, K, pose2d = _inputs()
    from cpu_shard_engine import cpu_optimizer_class
    from oracle import lbs_oracle as lo
    for k, fn in [('extra9', 'J_regressor_extra.npy'), ('h36m', 'J_regressor_h36m.npy'),
                  ('alphapose', 'SMPL_AlphaPose_Regressor_RMSprop_6.npy')]:
        np.save(os.path.join(tmp, fn), regs[k])
    model = lo.BodyModel(st, regs)
    opt = cpu_optimizer_class(model)(image_size=(W, H), num_frames=T, cam_K=K, device='cpu', smpl_model_parameters_path=tmp,
                                     smpl_data_struct=st, use_rasteriser=False,
                                     scene_update='none', use_graphs=False, shard_frames=True, **COEFS)
    opt.init_optimized_variables(pose2d, sp['poses_init'], sp['betas_init'], sp['valid'], num_iter=0)
    # a start away from the [0,0,1] of a skipped warm-up: the ground-truth translations, local slice per rank
    opt.engine.leaf('poses_T').copy_(torch.tensor(sp['trans_gt'][opt.first_frame:opt.last_frame]))
    opt.refresh_global_leaves()        # leaves edited by hand: re-gather what get_optimized_variables() serves (collective)
    ov0 = opt.get_optimized_variables()
    dl = torch.utils.data.DataLoader(_DS(sp, pose2d), batch_size=BATCH, shuffle=False)
    log = opt.fit(dl, num_iter=CYCLES, update_filters_every=31)
    opt.check_replicas()
    return opt, ov0, log, opt.get_optimized_variables()


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), MHHIP_CHECK_REPLICAS='1')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    tmp = os.path.join(out, 'regs%d' % rank)
    os.makedirs(tmp, exist_ok=True)
    opt, ov0, log, ov = _run_fit(tmp)
    torch.save(dict(ov0=ov0, ov=ov, log=log, first=opt.first_frame, last=opt.last_frame,
                    local_T=opt.poses_T.shape[0]), os.path.join(out, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize('world,bounds', [(2, [(0, 6), (6, 12)]), (3, [(0, 6), (6, 9), (9, 12)])])
def test_fit_on_several_ranks_matches_one(tmp_path, world, bounds):
    port = free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    one = os.path.join(str(tmp_path), 'one')
    os.makedirs(one)
    opt, ov0, log, ov = _run_fit(one)
    assert (opt.first_frame, opt.last_frame) == (0, T)
    r = [torch.load(os.path.join(str(tmp_path), 'rank%d.pt' % k), weights_only=False) for k in range(world)]
    assert [(x['first'], x['last']) for x in r] == bounds and r[0]['local_T'] == 6
    keys = ['scale_factor', 'poses_T', 'poses_smpl', 'betas_smpl', 'valid_smpl', 'min_z', 'max_z']
    for k in keys:
        # every rank returns the WHOLE sequence, identical across ranks ...
        for x in r[1:]:
            np.testing.assert_array_equal(r[0]['ov'][k], x['ov'][k], err_msg=k)
            np.testing.assert_array_equal(r[0]['ov0'][k], x['ov0'][k], err_msg=k)
        assert r[0]['ov'][k].shape == ov[k].shape, k
        # ... and equal to the single-process run (33 RMSprop steps incl. two with the filtered-vertex term)
        np.testing.assert_allclose(r[0]['ov0'][k], ov0[k], atol=1e-7, err_msg=k)
        # (the stand-in's LBS matmul blocks differently per batch size: last-ulp differences, amplified by RMSprop to 1e-4)
        np.testing.assert_allclose(r[0]['ov'][k], ov[k], atol=5e-4, err_msg=k)
    assert len(r[0]['log']) == CYCLES
    for c in range(CYCLES):
        for key in log[c]:
            np.testing.assert_allclose(r[0]['log'][c][key], log[c][key], rtol=2e-4, atol=1e-7, err_msg='%s cycle %d' % (key, c))
            np.testing.assert_allclose(r[-1]['log'][c][key], log[c][key], rtol=2e-4, atol=1e-7)
    assert log[32]['reg_filter_verts'] > 0 and log[30]['reg_filter_verts'] == 0
