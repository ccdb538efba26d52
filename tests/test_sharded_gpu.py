"""The frame-sharded driver on the real engine: two processes (both on cuda:0; collectives staged through gloo,
which only carries CPU tensors for all_gather / send / recv) must reproduce the single-process run of the full
nine-term cycle -- halos, shared-gradient all-reduce, one-euro hand-off, captured graphs with the halo exchange
between them.  The NCCL/RCCL transport itself is exercised by `bench.py --gpus N`."""
import os
import sys
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T, N, W, H, BATCH, CYCLES = 8, 2, 96, 54, 2, 4


def _paths():
    for p in (ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
        if p not in sys.path:
            sys.path.insert(0, p)


def _host_staged_dist():
    """torch.distributed with the tensor ops going through the host (gloo has no GPU all_gather / send / recv)"""
    shim = types.SimpleNamespace(**{k: getattr(dist, k) for k in dir(dist) if not k.startswith('__')})

    def all_gather(outs, t, group=None):
        tmp = [torch.empty(o.shape, dtype=o.dtype) for o in outs]
        dist.all_gather(tmp, t.detach().cpu(), group=group)
        for o, c in zip(outs, tmp):
            o.copy_(c)

    def all_reduce(t, op=dist.ReduceOp.SUM, group=None):
        c = t.detach().cpu()
        dist.all_reduce(c, op=op, group=group)
        t.copy_(c)

    def send(t, dst, group=None):
        dist.send(t.detach().cpu(), dst=dst, group=group)

    def recv(t, src, group=None):
        c = torch.empty(t.shape, dtype=t.dtype)
        dist.recv(c, src=src, group=group)
        t.copy_(c)

    class P2POp(object):
        def __init__(self, op, tensor, peer, group=None):
            self.op, self.tensor, self.peer, self.group = op, tensor, peer, group

    def batch_isend_irecv(ops):
        staged, reqs = [], []
        for o in ops:
            c = o.tensor.detach().cpu() if o.op is shim.isend else torch.empty(o.tensor.shape, dtype=o.tensor.dtype)
            staged.append(c)
            reqs.append((dist.isend if o.op is shim.isend else dist.irecv)(c, o.peer, group=o.group))
        for r in reqs:
            r.wait()
        for o, c in zip(ops, staged):
            if o.op is shim.irecv:
                o.tensor.copy_(c)
        return []

    shim.isend, shim.irecv = object(), object()
    shim.P2POp, shim.batch_isend_irecv = P2POp, batch_isend_irecv
    shim.all_gather, shim.all_reduce, shim.send, shim.recv = all_gather, all_reduce, send, recv
    return shim


def _build(f0, f1):
    _paths()
    from mhhip import synthetic, synthetic_seq, engine
    from mhhip.sequence import SequenceEngine
    import golden_inputs as gi
    struct = synthetic.make_smpl_struct(1)
    regs = synthetic.make_extra_regressors(1, struct)
    model = engine.BodyModel(struct, regs)
    K = synthetic.default_cam_K((W, H), 60.0)
    seq = synthetic_seq.make_sequence(model, N, T, (W, H), 41, cam_K=K, z_range=(2.6, 3.6))
    sl = slice(f0, f1)
    c = dict(gi.COEFS)
    e = SequenceEngine(model, (W, H), f1 - f0, N, K, None, c, batch_size=BATCH)
    sp = synthetic.make_sequence_params(N, T, 41)
    betas_ref = seq['betas_smpl'].mean(0)
    e.set_leaves(sp['trans_gt'][sl].astype(np.float32), seq['poses_smpl'][sl], betas_ref, np.ones(f1 - f0, np.float32),
                 6 * np.ones(f1 - f0, np.float32), np.zeros(N, np.float32))
    e.stage(seq['pose2d'][sl], seq['poses_smpl'][sl], seq['valid_smpl'][sl], betas_ref, seq['seg_mask'][sl], seq['depths'][sl])
    return e


def _backmasks():
    _paths()
    from mhhip import synthetic, synthetic_seq, engine
    struct = synthetic.make_smpl_struct(1)
    model = engine.BodyModel(struct, synthetic.make_extra_regressors(1, struct))
    K = synthetic.default_cam_K((W, H), 60.0)
    return synthetic_seq.make_sequence(model, N, T, (W, H), 41, cam_K=K, z_range=(2.6, 3.6))['backmasks']


def _run(sh, e):
    from mhhip.raster import RasterTerms
    raster = RasterTerms(e)
    for c in range(CYCLES):
        if c == 1:
            sh.update_filters()
        sh.cycle(c, raster=raster, graphs=True)
        sh.step()
    torch.cuda.synchronize()
    return sh.read_log(CYCLES)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    _paths()
    from mhhip import sharded
    sharded.dist = _host_staged_dist()
    f0, f1 = sharded.shard_bounds(T, world, BATCH)[rank]
    e = _build(f0, f1)
    sh = sharded.ShardedSequence(e, f0, T)
    log = _run(sh, e)
    # pixel-sharded scene aggregation: every rank ends with the scene of the WHOLE sequence
    import numpy as np_
    from mhhip import synthetic as syn_, synthetic_seq as sseq_
    sh.scene_setup(_backmasks()[f0:f1])
    sh.scene_update()
    sh.scene_swap()
    depth, mask, pts = e.scene_device_result()
    torch.save(dict(params=e.params.cpu(), log=log, f0=f0, f1=f1, scene_depth=depth, scene_mask=mask, npts=pts.shape[0]),
               os.path.join(out, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_ranks_on_the_device_match_one(tmp_path):
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    _paths()
    from mhhip import sharded
    e = _build(0, T)
    sh = sharded.ShardedSequence(e, 0, T)
    log = _run(sh, e)
    r = [torch.load(os.path.join(str(tmp_path), 'rank%d.pt' % k), weights_only=False) for k in range(2)]
    assert (r[0]['f0'], r[0]['f1'], r[1]['f0'], r[1]['f1']) == (0, 4, 4, 8)
    for name in ['poses_T', 'poses_smpl', 'zmin_lin', 'zmax_lin']:
        full = e.leaf(name).cpu().numpy()
        for k in range(2):
            ek = _build(r[k]['f0'], r[k]['f1'])
            got = ek.leaf(name, r[k]['params'].to(ek.dev)).cpu().numpy()
            want = full[r[k]['f0']:r[k]['f1']]
            # float atomics in the raster gradients: tight but not bit-exact
            np.testing.assert_allclose(got, want, atol=3e-4 * max(1.0, np.abs(want).max()), err_msg=name)
    for name in ['betas', 'xscale']:
        for k in range(2):
            ek = _build(r[k]['f0'], r[k]['f1'])
            np.testing.assert_allclose(ek.leaf(name, r[k]['params'].to(ek.dev)).cpu().numpy(), e.leaf(name).cpu().numpy(), atol=3e-4,
                                       err_msg=name)
    for c in range(CYCLES):
        for key in log[c]:
            np.testing.assert_allclose(r[0]['log'][c][key], log[c][key], rtol=2e-3, atol=1e-6, err_msg='%s cycle %d' % (key, c))
            np.testing.assert_allclose(r[1]['log'][c][key], log[c][key], rtol=2e-3, atol=1e-6)
    assert log[2]['reg_filter_verts'] > 0 and log[0]['loss_depth'] > 0
    # scene of the whole sequence from the single process (same leaves up to the atomics noise of the run above)
    sh.scene_setup(_backmasks())
    sh.scene_update()
    sh.scene_swap()
    depth, mask, pts = e.scene_device_result()
    for k in range(2):
        np.testing.assert_array_equal(r[k]['scene_mask'], mask)
        bad = np.abs(r[k]['scene_depth'] - depth) > 2e-3 * np.maximum(1.0, np.abs(depth))
        assert bad.mean() < 0.01, bad.sum()
        assert abs(r[k]['npts'] - pts.shape[0]) == 0
