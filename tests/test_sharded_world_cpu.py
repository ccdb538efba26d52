"""The frame-sharded driver (mhhip/sharded.py) at world sizes 3 and 8 on the gloo backend (CPU stand-in engine,
tests/cpu_shard_engine.py): what the 2-rank tests cannot reach --

* middle ranks with BOTH neighbours (two poses_T halos, two vertex halos, both filtered-vertex halos),
* a one-euro hand-off over world-1 hops,
* uneven shards (22 frames in batches of 2: 8/8/6 frames on 3 ranks, 4/4/4/2/2/2/2/2 on 8),
* the pixel-sharded scene aggregation (depth median over ALL frames per pixel slice, padded frames, medians
  all-gathered, post-processing on every rank) and the pixel-sharded scene image,
* a process group that does not start at global rank 0 (point-to-point peers are GLOBAL ranks),

all against the single-process run of the same code with world size 1."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from netutil import free_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T, N, W, H, BATCH, CYCLES = 22, 2, 48, 32, 2, 4


def _setup_paths():
    for p in (ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
        if p not in sys.path:
            sys.path.insert(0, p)


def _inputs():
    _setup_paths()
    import test_sharded_cpu as base
    base.T = T                              # same generator, longer sequence
    model, sp, K, pose2d = base._inputs()
    rng = np.random.RandomState(9)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    depths = np.clip(0.5 + 0.3 * np.sin(xs / 7.0)[None] + 0.1 * rng.rand(T, H, W), 0.02, 1).astype(np.float32)
    back = (rng.rand(T, H, W) > 0.35).astype(np.int64)
    back[:, 10:16, 20:26] = 0               # never seen: filled by the post-processing
    back[5:, 3, 3] = 0                      # seen by the first rank's frames only
    images = rng.randint(0, 256, (T, H, W, 3)).astype(np.uint8)
    return base, model, sp, K, pose2d, depths, back, images


def _run_shard(group, grank, world):
    base, model, sp, K, pose2d, depths, back, images = _inputs()
    from mhhip.sharded import ShardedSequence, shard_bounds
    f0, f1 = shard_bounds(T, world, BATCH)[grank]
    base.T = T
    e = base._make_engine(model, sp, K, pose2d, f0, f1)
    e.set_images(depths[f0:f1])
    e.leaf('zmin_lin').copy_(torch.linspace(0.8, 1.3, T)[f0:f1])
    e.leaf('zmax_lin').copy_(torch.linspace(5.0, 8.0, T)[f0:f1])
    sh = ShardedSequence(e, f0, T, group=group)
    assert sh.world == world and sh.rank == grank
    base.CYCLES = CYCLES
    log = base._run(sh, e)
    sh.scene_setup(back[f0:f1])
    sh.scene_update()
    sh.scene_swap()
    depth, mask, pts = e.scene_device_result()
    img, imask = sh.scene_image(images[f0:f1])
    # the one-euro hand-off in chunks (round 6: rank k + 1 starts on chunk j while rank k scans chunk j + 1) against the
    # one-piece hand-off: the same bits
    xs = torch.from_numpy(np.random.RandomState(77).randn(T, 37).astype(np.float32))[f0:f1].contiguous()
    scan_same = all(torch.equal(sh._scan(xs, 0.01, 0.5, chunks=1), sh._scan(xs, 0.01, 0.5, chunks=ch)) for ch in (2, 5, 37, 64))
    return dict(scan_same=scan_same, params=e.params.clone(), log=log, f0=f0, f1=f1, pT_filt=e.pT_filt, vf=e.verts_filt[:, :, ::97],
                scene_depth=depth, scene_mask=mask, npts=pts.shape[0], img=img, imask=imask)


def _worker(rank, nproc, port, out, offset):
    """offset > 0: the sharding group is ranks offset..nproc-1 of the world (rank 0.. offset-1 only join the barriers)"""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=nproc)
    torch.set_num_threads(1)
    group = None
    if offset:
        group = dist.new_group(ranks=list(range(offset, nproc)))
    if rank >= offset:
        res = _run_shard(group, rank - offset, nproc - offset)
        torch.save(res, os.path.join(out, 'rank%d.pt' % (rank - offset)))
    dist.barrier()
    dist.destroy_process_group()


def _check(tmp_path, world, want_bounds):
    one = _run_shard(None, 0, 1)
    r = [torch.load(os.path.join(str(tmp_path), 'rank%d.pt' % k), weights_only=False) for k in range(world)]
    assert [(x['f0'], x['f1']) for x in r] == want_bounds
    assert all(x['scan_same'] for x in r) and one['scan_same']
    base = sys.modules['test_sharded_cpu']
    _, model, sp, K, pose2d, _, _, _ = _inputs()
    ref = base._make_engine(model, sp, K, pose2d, 0, T)
    for name in ['poses_T', 'poses_smpl', 'zmin_lin', 'zmax_lin']:
        full = ref.leaf(name, one['params']).numpy()
        for k in range(world):
            ek = base._make_engine(model, sp, K, pose2d, r[k]['f0'], r[k]['f1'])
            np.testing.assert_allclose(ek.leaf(name, r[k]['params']).numpy(), full[r[k]['f0']:r[k]['f1']], atol=3e-6,
                                       err_msg='%s rank %d' % (name, k))
    for name in ['betas', 'xscale']:
        for k in range(world):
            ek = base._make_engine(model, sp, K, pose2d, r[k]['f0'], r[k]['f1'])
            got = ek.leaf(name, r[k]['params']).numpy()
            np.testing.assert_allclose(got, ref.leaf(name, one['params']).numpy(), atol=3e-6, err_msg=name)
            np.testing.assert_array_equal(got, base._make_engine(model, sp, K, pose2d, r[0]['f0'], r[0]['f1']).leaf(name, r[0]['params']).numpy())
    # the filter recurrence continued over world-1 hand-offs
    np.testing.assert_allclose(torch.cat([x['pT_filt'] for x in r]).numpy(), one['pT_filt'].numpy(), atol=1e-6)
    np.testing.assert_allclose(torch.cat([x['vf'] for x in r]).numpy(), one['vf'].numpy(), atol=1e-5)
    for c in range(CYCLES):
        for key in one['log'][c]:
            for k in range(world):
                np.testing.assert_allclose(r[k]['log'][c][key], one['log'][c][key], rtol=2e-5, atol=1e-7,
                                           err_msg='%s cycle %d rank %d' % (key, c, k))
    assert one['log'][2]['reg_filter_verts'] > 0 and one['log'][0]['reg_vel'] > 0
    # scene: medians over the frames of ALL ranks -- identical on every rank and equal to the single process, bit for bit
    # up to the leaves (the depth-range leaves moved identically to 3e-6 above)
    for k in range(world):
        np.testing.assert_array_equal(r[k]['scene_mask'], one['scene_mask'])
        np.testing.assert_allclose(r[k]['scene_depth'], one['scene_depth'], rtol=2e-5)
        assert r[k]['npts'] == one['npts']
        np.testing.assert_array_equal(r[k]['img'], one['img'])
        np.testing.assert_array_equal(r[k]['imask'], one['imask'])
    assert one['scene_mask'][3, 3] and not one['scene_mask'][12, 22] and one['imask'].min() == 1


@pytest.mark.timeout(900)
def test_three_ranks_uneven_shards(tmp_path):
    port = free_port()
    mp.spawn(_worker, args=(3, port, str(tmp_path), 0), nprocs=3, join=True)
    _check(tmp_path, 3, [(0, 8), (8, 16), (16, 22)])


@pytest.mark.timeout(1200)
def test_eight_ranks(tmp_path):
    port = free_port()
    mp.spawn(_worker, args=(8, port, str(tmp_path), 0), nprocs=8, join=True)
    _check(tmp_path, 8, [(0, 4), (4, 8), (8, 12), (12, 14), (14, 16), (16, 18), (18, 20), (20, 22)])


@pytest.mark.timeout(900)
def test_subgroup_not_starting_at_global_rank_zero(tmp_path):
    """4 processes, the sequence sharded over the group of global ranks 1..3: halos and filter state must reach the
    group neighbours (P2POp / send / recv address GLOBAL ranks)"""
    port = free_port()
    mp.spawn(_worker, args=(4, port, str(tmp_path), 1), nprocs=4, join=True)
    _check(tmp_path, 3, [(0, 8), (8, 16), (16, 22)])
