"""Frame-sharded optimisation over the GPUs of one node: one process per GPU, frames split in
contiguous blocks (block boundaries at multiples of the batch size so the in-batch terms are
unchanged), per-frame leaves and inputs local to their rank.

Per cycle the ranks exchange, over ``torch.distributed`` (backend "nccl" == RCCL on ROCm, xGMI):

  * ONE all-reduce (sum) of the gradient tail of the shared leaves ``betas (N,10) | xscale (N)``
    -- 11*N floats, latency-bound -- after which every rank applies the identical RMSprop step to
    its replica of those leaves;
  * the boundary frame of ``poses_T`` with both neighbours (velocity term, optimizer.py:560), and,
    once the one-euro filters exist (cycle >= 50), the boundary frame's vertices (filtered-vertex
    term, optimizer.py:571-573) -- gathered with one small all_gather each;
  * every 25 cycles the one-euro filter state (filtered value + filtered derivative of the last
    local frame) is handed rank k -> k+1, because the filter is sequential in time
    (optimizer.py:664-675).

The reference has no distributed path at all; this file is new functionality and is
parity-tested against the single-process run.  The compute engine is duck-typed (``SequenceEngine``
on the GPU; the CPU tests plug in a torch-CPU stand-in with the same methods).
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(num_frames, world, batch_size):
    """Contiguous frame ranges per rank, boundaries at multiples of the batch size."""
    nb = (num_frames + batch_size - 1) // batch_size
    per, extra = divmod(nb, world)
    bounds, s = [], 0
    for r in range(world):
        n = per + (1 if r < extra else 0)
        e = min(num_frames, s + n * batch_size)
        bounds.append((s, e))
        s = e
    return bounds


class ShardedSequence(object):
    def __init__(self, engine, first_frame, total_frames, group=None):
        self.e = engine
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.first_frame = int(first_frame)
        self.total_frames = int(total_frames)
        self.is_first = self.first_frame == 0
        self.is_last = self.first_frame + engine.T >= self.total_frames
        self._vf_halo = None
        self._hbuf = {}

    # -- neighbour exchange: every rank contributes its first and last frame ------------------------
    def _gather_boundaries(self, x):
        """x (T_local, E...) -> (prev_rank_last, next_rank_first) or None at the sequence ends."""
        if self.world == 1:
            return None, None
        mine = torch.stack([x[0], x[-1]]).contiguous()
        out = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(out, mine, group=self.group)
        prev = None if self.is_first else out[self.rank - 1][1].contiguous()
        nxt = None if self.is_last else out[self.rank + 1][0].contiguous()
        return prev, nxt

    def _static(self, name, t):
        """halo tensors live at fixed addresses so that captured launches can read them"""
        if t is None:
            return None
        buf = self._hbuf.get(name)
        if buf is None or buf.shape != t.shape:
            buf = torch.empty_like(t)
            self._hbuf[name] = buf
        buf.copy_(t)
        return buf

    def cycle(self, row, raster=None, graphs=False):
        e = self.e
        if self.world == 1:
            e.halo = None
            if graphs:
                e.cycle_graphed(row, raster=raster)
            else:
                e.cycle_begin()
                e.cycle_finish(row, raster=raster)
            return
        halo = {}
        pp, pn = self._gather_boundaries(e.leaf('poses_T'))
        halo['pT_prev'], halo['pT_next'] = self._static('pT_prev', pp), self._static('pT_next', pn)
        if graphs:
            e.replay(('begin',), e.cycle_begin)
        else:
            e.cycle_begin()
        if e.verts_filt is not None and e.pT_filt is not None:
            vp, vn = self._gather_boundaries(e.verts.view(e.T, -1))
            halo['v_prev'], halo['v_next'] = self._static('v_prev', vp), self._static('v_next', vn)
            halo['vf_prev'], halo['vf_next'] = self._vf_halo
        e.halo = halo
        if graphs:
            e.replay(('finish',) + e._graph_key(raster), lambda: e.cycle_finish(None, raster=raster))
            e.log[row].copy_(e.tmp_log)
        else:
            e.cycle_finish(row, raster=raster)
        dist.all_reduce(e.grads[e.shared_lo:], op=dist.ReduceOp.SUM, group=self.group)

    def step(self, lr=None):
        """lr given: host-side schedule; lr None: the device-resident schedule (graph friendly)"""
        if lr is None:
            self.e.step_dev()
        else:
            self.e.step(lr)

    # -- one-euro filters with the state handed down the ranks (optimizer.py:383-392) ----------------
    def _scan(self, x, c, b):
        e = self.e
        state = None
        if not self.is_first:
            E = x.numel() // x.shape[0]
            xp = torch.empty(E, dtype=torch.float32, device=x.device)
            dxp = torch.empty(E, dtype=torch.float32, device=x.device)
            dist.recv(xp, src=self.rank - 1, group=self.group)
            dist.recv(dxp, src=self.rank - 1, group=self.group)
            state = (xp, dxp)
        y, out = e.one_euro_shard(x, c, b, self.first_frame, state)
        if not self.is_last:
            dist.send(out[0], dst=self.rank + 1, group=self.group)
            dist.send(out[1], dst=self.rank + 1, group=self.group)
        return y

    def update_filters(self, c1=0.01, b1=0.02, c2=0.001, b2=0.5):
        e = self.e
        pf = self._scan(e.leaf('poses_T'), c1, b1)
        e.forward()
        vf = self._scan(e.verts.view(e.T, -1), c2, b2).view(e.verts.shape[0] // e.N, e.N, -1, 3)
        if hasattr(e, 'set_filters'):
            e.set_filters(pf, vf)            # fixed addresses: the captured cycle graphs read these buffers
        else:
            e.pT_filt, e.verts_filt = pf, vf
        a, b = self._gather_boundaries(e.verts_filt.view(e.T, -1))
        self._vf_halo = (self._static('vf_prev', a), self._static('vf_next', b))

    # -- logs: raw sums are all-reduced once, at the end ----------------------------------------------
    def read_log(self, rows):
        e = self.e
        raw = e.log[:rows].clone()
        if self.world > 1:
            dist.all_reduce(raw, op=dist.ReduceOp.SUM, group=self.group)
            raw[:, 10:12] /= self.world          # the scale regularisers are replicated, not sharded
        nb = torch.tensor([float(e.nbatches)], device=raw.device)
        if self.world > 1:
            dist.all_reduce(nb, op=dist.ReduceOp.SUM, group=self.group)
        keep, e.log = e.log, raw
        try:
            return e.read_log(rows, nbatches_total=float(nb.item()))
        finally:
            e.log = keep
