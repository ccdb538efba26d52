"""The RCCL backend itself on the GPU box: a one-rank ``nccl`` process group initialised the way bench.py / INTEGRATION.md do
(``device_id=``), running the collectives the frame-sharded driver uses (all_reduce on a tail view, all_gather, broadcast, a
sub-group) on device tensors.  One GPU per box: this is as much of the RCCL path as can execute here -- the N-rank logic is
covered by the gloo tests and by tests/test_bench_multirank_gpu.py (N processes on one device, host-staged collectives)."""
import os
import subprocess
import sys

import pytest

from netutil import free_port

pytestmark = pytest.mark.gpu

SCRIPT = r'''
import os, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
g = torch.arange(64, dtype=torch.float32, device='cuda:0')
dist.all_reduce(g[22:], op=dist.ReduceOp.SUM)                 # the betas|xscale tail of the gradient buffer
outs = [torch.empty(5, 7, device='cuda:0')]
dist.all_gather(outs, torch.ones(5, 7, device='cuda:0'))
t = torch.full((3,), 2.0, device='cuda:0')
dist.broadcast(t, src=0)
sub = dist.new_group([0])
dist.all_reduce(t, op=dist.ReduceOp.MAX, group=sub)
dist.barrier()
torch.cuda.synchronize()
assert float(g.sum()) == 2016.0 and float(outs[0].sum()) == 35.0 and float(t.sum()) == 6.0
print('RCCL_OK', dist.get_backend())
dist.destroy_process_group()
'''


def test_rccl_initialises_and_runs_the_drivers_collectives():
    port = free_port()                                # (another job on the box may hold a fixed one)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-c', SCRIPT], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'RCCL_OK nccl' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


GRAPH_SCRIPT = r'''
import os, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
# the frame-sharded driver's message of a cycle: [gradient tail | every rank's boundary leaves], summed over the ranks
ar = torch.zeros(44 + 600, device='cuda:0')
src = torch.arange(644, dtype=torch.float32, device='cuda:0')
dist.all_reduce(ar)                                           # warm-up outside the capture (communicator set-up)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        ar.copy_(src)                                         # "pack"
        dist.all_reduce(ar, op=dist.ReduceOp.SUM)
        ar.mul_(2.0)                                          # "unpack + shared step"
except Exception as ex:
    print('CAPTURE_UNSUPPORTED', type(ex).__name__, str(ex)[:200])
    raise SystemExit(0)
for k in range(3):
    src.add_(1.0)
    g.replay()
torch.cuda.synchronize()
want = 2.0 * (torch.arange(644, dtype=torch.float32, device='cuda:0') + 3.0)
assert torch.equal(ar, want), float((ar - want).abs().max())
print('RCCL_GRAPH_OK')
dist.destroy_process_group()
'''


def test_the_cycles_all_reduce_replays_from_a_captured_graph():
    """can the RCCL all-reduce sit INSIDE a captured graph?  (The driver issues it eagerly between two
    RMSprop launches today: one replay per cycle, one collective, two small launches.)  One-rank group -- as much of RCCL as
    a one-GPU box runs: pack -> all_reduce -> unpack captured once and replayed three times on changing inputs."""
    port = free_port()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-c', GRAPH_SCRIPT], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    if 'CAPTURE_UNSUPPORTED' in r.stdout:
        pytest.skip('this RCCL / PyTorch build does not capture collectives: ' + r.stdout.strip()[-200:])
    assert 'RCCL_GRAPH_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
