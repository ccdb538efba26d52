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
