"""``bench.py`` in its multi-rank forms, executed end to end on ONE device: N processes under ``torch.distributed.run``
exactly as the driver launches the scaling bench, with ``--backend gloo --one-device`` (collectives staged through the
host, all ranks on cuda:0 -- RCCL refuses two ranks per device and this box has one GPU).  Everything except the RCCL
transport itself runs: the drop-in's own sharding of ONE contiguous sequence, uneven blocks, middle ranks with two
neighbours, the multi-hop one-euro hand-off, halos between the captured graphs, the all-reduce per cycle, replica check,
weak (default / --config c4) and strong (--strong) accounting of the JSON line.  Toy sizes; the numbers are not results."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(nproc, extra, port):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', MHHIP_CHECK_REPLICAS='1', OMP_NUM_THREADS='2')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', str(nproc), '--backend', 'gloo', '--one-device',
           '--steps', '30', '--warmup', '2', '--presteps', '4', '--image', '96x54', '--humans', '2'] + extra
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(1800)
def test_gpus_flag_alone_launches_the_ranks():
    """``python bench.py --gpus 2`` WITHOUT a launcher must become two ranks by itself (it used to run one process
    and print n_gpus 1), and the line must carry what the process group saw"""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='2')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--one-device', '--steps', '12', '--warmup', '2',
           '--presteps', '4', '--image', '96x54', '--humans', '2', '--frames', '20']
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][0])
    assert out['n_gpus'] == 2 and out['config']['frames'] == 40
    assert out['rccl']['ranks'] == 2 and out['rccl']['allreduce_of_ones'] == 2 and out['rccl']['one_device_dry_run'] is True


@pytest.mark.timeout(1800)
def test_weak_scaling_form_on_three_ranks():
    out = _run(3, ['--frames', '20'], 29811)
    assert out['n_gpus'] == 3 and out['scaling'] == 'weak' and out['steps'] == 30
    assert out['config']['frames'] == 60 and out['config']['frames_per_gpu'] == 20
    assert out['value'] > 0 and abs(out['value'] - 3 * 30 / out['timed_region_s']) < 1e-2 * out['value']
    assert out['filter_updates_in_timed_region'] >= 1 and 'DRY RUN' in out['backend']
    assert all(v == v for v in out['loss_first_cycle'].values())          # finite log, reduced over the ranks


@pytest.mark.timeout(1800)
def test_c4_form_on_eight_ranks():
    out = _run(8, ['--config', 'c4', '--frames', '10'], 29823)
    assert out['n_gpus'] == 8 and out['config']['frames'] == 80 and out['scaling'] == 'weak'
    assert out['loss_first_cycle']['reg_filter_verts'] > 0 and out['loss_first_cycle']['reg_contact'] > 0


@pytest.mark.timeout(1800)
def test_strong_scaling_form_uneven_shards():
    """--strong: a FIXED job (here 50 frames in batches of 10 over 3 ranks: blocks of 20 / 20 / 10) with the injected
    200 000-point cloud; value counts whole-job iterations, not x ranks"""
    out = _run(3, ['--strong', '--frames', '50', '--humans', '3'], 29837)
    assert out['n_gpus'] == 3 and out['scaling'] == 'strong' and out['config']['frames'] == 50 and out['config']['humans'] == 3
    assert abs(out['value'] - 30 / out['timed_region_s']) < 1e-2 * out['value']
    assert out['loss_first_cycle']['reg_contact'] > 0


@pytest.mark.timeout(3000)
def test_c4_at_full_size_eight_ranks_equal_one_process(tmp_path):
    """BASELINE C4 exactly -- 4 humans x 2000 frames at 240x135, 250 frames per rank -- as eight ranks on this one device
    (gloo, collectives through the host: everything but the RCCL transport) against the SAME 4 x 2000 sequence in one
    process on the same GPU: the nine-term stack, kept face lists, the projection epilogue, one graph and one all-reduce
    per cycle on the eight ranks.  VERDICT r03: the multi-rank maths had only ever run at toy sizes.

    (1) THE MATHS: the gradient of every leaf of one unstepped cycle at identical variables (deterministic scatter).
    (2) THE TRAJECTORIES, 18 cycles and a one-euro filter update later, against what two runs of the one-process job part by.

    Until round 4 this test had to repeat the eight-rank run: a few frames -- different ones every time -- came out with
    different selection keys.  It was not the eight processes: the frame-sharded cycle skins its neighbours' boundary frames
    in the side branch, beside the selection kernel, and hipcc had packed part of that kernel's fp32 arithmetic
    (``v_pk_*_f32``), which returns wrong values while another wave's matrix instructions are in flight on the same SIMD
    (mhhip/build.py, tests/test_corunner_gpu.py).  Without packed instructions the FIRST run must be
    right in every entry."""
    import numpy as np
    common = ['--steps', '12', '--warmup', '2', '--presteps', '4', '--no-cpu-baseline', '--no-fit']
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='2')

    def run_one(path):
        p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--frames', '2000', '--dump-leaves', path] + common,
                           cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1400)
        assert p.returncode == 0, p.stderr[-3000:]
        return np.load(path)

    def run_eight(path, port):
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--backend', 'gloo', '--one-device', '--config', 'c4',
               '--dump-leaves', path] + common
        p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1400)
        assert p.returncode == 0, p.stderr[-3000:]
        out = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][0])
        assert out['n_gpus'] == 8 and out['config']['frames'] == 2000 and out['config']['frames_per_gpu'] == 250
        return np.load(path)

    a = run_one(str(tmp_path / 'one.npz'))
    c = run_one(str(tmp_path / 'again.npz'))                  # the same one-process run a second time: what any two runs part by
    GRADS = ('grad0_poses_T', 'grad0_poses_smpl', 'grad0_zmin_lin', 'grad0_zmax_lin', 'grad0_tail')
    for k in GRADS:
        assert np.array_equal(a[k], c[k]), k                   # one process: bit-identical from run to run
    b = run_eight(str(tmp_path / 'eight.npz'), 29851)
    # the LBS backward cuts the vertices into 500 / (groups of 32 bodies) chunks and adds the chunk sums in order: 2 chunks for
    # the one process' 250 groups, 16 for a rank's 32, so sums of 6890 cancelling terms round differently; the shared tail is
    # additionally summed per rank and then across ranks
    worst = {k: float(np.abs(a[k] - b[k]).max() / np.abs(a[k]).max()) for k in GRADS}
    print('first-cycle gradients, max |eight ranks - one process| / largest entry: %s' % ', '.join('%s %.1e' % (k[6:], v) for k, v in worst.items()))
    assert max(worst.values()) <= 2e-5, 'first-cycle gradients of the eight-rank run differ from the one-process run: %s' % worst
    for k in ('poses_T', 'poses_smpl', 'betas', 'zmin_lin', 'zmax_lin', 'xscale'):
        assert a[k].shape == b[k].shape, k
        d, d1 = np.abs(a[k] - b[k]), np.abs(a[k] - c[k])
        print('%-10s |eight ranks - one process|: median %.2e  99%% %.2e  max %.2e   (one process twice: median %.2e  99%% %.2e  max %.2e)'
              % (k, np.median(d), np.percentile(d, 99), d.max(), np.median(d1), np.percentile(d1, 99), d1.max()))
        # RMSprop's first steps are sign-like (g / sqrt(0.5 g^2)): an entry whose gradient is rounding noise moves by +-lr whichever
        # way the noise points, so single entries part by centimetres in ANY two runs (the float atomics of the production scatter
        # are enough: the right-hand columns); the bulk must not, and the tail must look like that of two one-process runs
        # (18 cycles of a chaotic trajectory: a sanity bound, the statement of this test is the gradient check above; the
        # shared leaves have 4-40 entries, their "median" is one of them)
        assert np.median(d) <= max(5e-4, 10.0 * float(np.median(d1))), (k, float(np.median(d)), float(np.median(d1)))
        # (the spread of two one-process runs itself varies tenfold from run to run -- betas: 2.4e-4 one time, 2.9e-3 the next --
        # so the tail is held against the larger of four times this run's spread and the largest spread seen)
        assert np.percentile(d, 99) <= max(2e-2, 4.0 * float(np.percentile(d1, 99))), (k, float(np.percentile(d, 99)), float(np.percentile(d1, 99)))
