"""world_size-2 (gloo, CPU) test of the frame-sharded driver mhhip/sharded.py: two ranks owning half
the frames each must reproduce the single-process run -- shared-gradient all-reduce, poses_T /
vertex halos, one-euro state hand-off, log reduction."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from netutil import free_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

T, N, W, H, BATCH, CYCLES = 8, 2, 48, 32, 2, 4
COEFS = dict(proj2d=1.0, reg_poses=0.002, reg_scales=1e-4, reg_velocity=0.05, reg_verts_filter=0.002)


def _inputs():
    for p in (ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
        if p not in sys.path:
            sys.path.insert(0, p)
    from mhhip import synthetic
    from oracle import lbs_oracle as lo
    st = synthetic.make_smpl_struct(1)
    regs = synthetic.make_extra_regressors(1, st)
    model = lo.BodyModel(st, regs)
    sp = synthetic.make_sequence_params(N, T, 77)
    rng = np.random.RandomState(5)
    K = synthetic.default_cam_K((W, H), 60.0)
    pose2d = np.zeros((T, N, 17, 3), np.float32)
    pose2d[..., 0] = rng.uniform(0, W, (T, N, 17)); pose2d[..., 1] = rng.uniform(0, H, (T, N, 17))
    pose2d[..., 2] = rng.uniform(0.3, 1.0, (T, N, 17))
    return model, sp, K, pose2d


def _make_engine(model, sp, K, pose2d, f0, f1):
    from cpu_shard_engine import CpuShardEngine
    sl = slice(f0, f1)
    betas_ref = sp['betas_init'].mean(0)
    e = CpuShardEngine(model, (W, H), f1 - f0, N, K, COEFS, BATCH, pose2d[sl], sp['poses_init'][sl], sp['valid'][sl], betas_ref)
    e.leaf('poses_T').copy_(torch.tensor(sp['trans_gt'][sl]))
    e.leaf('poses_smpl').copy_(torch.tensor(sp['poses_init'][sl]))
    e.leaf('betas').copy_(torch.tensor(betas_ref))
    e.leaf('zmin_lin').fill_(1.0); e.leaf('zmax_lin').fill_(8.0)
    return e


def _run(sh, e):
    lr = 0.01
    sh.refresh_halo()                   # collective, every rank: the cycles issue no hidden one
    for c in range(CYCLES):
        if c == 1:                      # filters come alive (cycle 50 in the reference schedule)
            sh.update_filters()
        sh.cycle(c)
        sh.step(lr)
        lr *= 0.99
    return sh.read_log(CYCLES)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    model, sp, K, pose2d = _inputs()
    from mhhip.sharded import ShardedSequence, shard_bounds
    f0, f1 = shard_bounds(T, world, BATCH)[rank]
    e = _make_engine(model, sp, K, pose2d, f0, f1)
    sh = ShardedSequence(e, f0, T)
    # a cycle on stale halos raises instead of starting a hidden collective (a cycle run on some ranks only would
    # otherwise hang the others); the device-style schedule of step(lr=None) restarts when the leaves are set from outside
    try:
        sh.cycle(0)
        raise AssertionError('a frame-sharded cycle on stale halos must raise')
    except RuntimeError as ex:
        assert 'refresh_halo' in str(ex)
    sh._lr32 = np.float32(0.005)
    sh.leaves_changed()
    assert sh._lr32 is None
    log = _run(sh, e)
    torch.save(dict(params=e.params.clone(), log=log, f0=f0, f1=f1, pT_filt=e.pT_filt, vf=e.verts_filt[:, :, ::97]),
               os.path.join(out, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_match_single_process(tmp_path):
    port = free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    # single process reference
    model, sp, K, pose2d = _inputs()
    from mhhip.sharded import ShardedSequence
    e = _make_engine(model, sp, K, pose2d, 0, T)
    sh = ShardedSequence(e, 0, T)
    log = _run(sh, e)
    r = [torch.load(os.path.join(str(tmp_path), 'rank%d.pt' % k), weights_only=False) for k in range(2)]
    assert (r[0]['f0'], r[0]['f1'], r[1]['f0'], r[1]['f1']) == (0, 4, 4, 8)
    for name in ['poses_T', 'poses_smpl', 'zmin_lin', 'zmax_lin']:
        full = e.leaf(name).numpy()
        for k in range(2):
            ek = _make_engine(model, sp, K, pose2d, r[k]['f0'], r[k]['f1'])
            got = ek.leaf(name, r[k]['params']).numpy()
            np.testing.assert_allclose(got, full[r[k]['f0']:r[k]['f1']], atol=2e-6, err_msg=name)
    for name in ['betas', 'xscale']:                      # shared leaves: identical replicas == single run
        for k in range(2):
            ek = _make_engine(model, sp, K, pose2d, r[k]['f0'], r[k]['f1'])
            np.testing.assert_allclose(ek.leaf(name, r[k]['params']).numpy(), e.leaf(name).numpy(), atol=2e-6, err_msg=name)
    # one-euro state hand-off: the second rank continues the first rank's filter exactly
    pf = torch.cat([r[0]['pT_filt'], r[1]['pT_filt']]).numpy()
    np.testing.assert_allclose(pf, e.pT_filt.numpy(), atol=1e-6)
    vf = torch.cat([r[0]['vf'], r[1]['vf']]).numpy()
    np.testing.assert_allclose(vf, e.verts_filt[:, :, ::97].numpy(), atol=5e-6)   # LBS matmul blocking differs with the batch
    for c in range(CYCLES):
        for key in log[c]:
            np.testing.assert_allclose(r[0]['log'][c][key], log[c][key], rtol=1e-5, atol=1e-7, err_msg='%s cycle %d' % (key, c))
            np.testing.assert_allclose(r[1]['log'][c][key], log[c][key], rtol=1e-5, atol=1e-7)
    assert log[2]['reg_filter_verts'] > 0
