"""TEST INFRASTRUCTURE (moved out of the product package in round 4): ``torch.distributed`` with every tensor operation
staged through host memory.

The frame-sharded driver (mhhip/sharded.py) talks to ``torch.distributed`` directly; on the 8-GPU node that is RCCL over
xGMI (backend "nccl").  The ``gloo`` backend only carries device tensors for broadcast / all_reduce, and RCCL refuses
two ranks on one device, so a dry run of the multi-rank path on a single GPU (``bench.py --backend gloo``; the
2-process tests of tests/test_sharded_gpu.py) needs the collectives and the point-to-point calls staged through the
host.  Plumbing only: bytes are copied, nothing is computed."""
import types

import torch
import torch.distributed as dist


def host_staged():
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

    def broadcast(t, src, group=None):
        c = t.detach().cpu()
        dist.broadcast(c, src=src, group=group)
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
    shim.all_gather, shim.all_reduce, shim.send, shim.recv, shim.broadcast = all_gather, all_reduce, send, recv, broadcast
    return shim


def install():
    """route the sharded driver and the drop-in optimiser through the host-staged operations; returns the shim"""
    from mhhip import sharded
    import mhmocap.optimizer as mo
    shim = host_staged()
    sharded.dist = shim
    mo.dist = shim
    return shim
