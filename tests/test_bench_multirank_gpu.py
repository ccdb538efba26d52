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
