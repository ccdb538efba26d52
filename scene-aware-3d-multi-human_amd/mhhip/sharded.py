"""Frame-sharded optimisation over the GPUs of one node: one process per GPU, frames split in
contiguous blocks (block boundaries at multiples of the batch size so the in-batch terms are
unchanged), per-frame leaves and inputs local to their rank.

Per cycle the ranks exchange, over ``torch.distributed`` (backend "nccl" == RCCL on ROCm, xGMI):

  * ONE all-reduce (sum) of the gradient tail of the shared leaves ``betas (N,10) | xscale (N)``
    -- 11*N floats, latency-bound -- after which every rank applies the identical RMSprop step to
    its replica of those leaves;
  * the boundary frame of ``poses_T`` with both neighbours (velocity term, optimizer.py:560), and,
    once the one-euro filters exist (cycle >= 50), the boundary frame's vertices (filtered-vertex
    term, optimizer.py:571-573) -- exchanged point-to-point with the two neighbours;
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
    def __init__(self, engine, first_frame, total_frames, group=None, enabled=True):
        """enabled=False: single-process behaviour even when torch.distributed is initialised (sharding is opt-in)"""
        self.e = engine
        self.group = group
        on = enabled and dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if on else 0
        self.world = dist.get_world_size(group) if on else 1
        self.first_frame = int(first_frame)
        self.total_frames = int(total_frames)
        self.is_first = self.first_frame == 0
        self.is_last = self.first_frame + engine.T >= self.total_frames
        self._vf_halo = None
        self._hbuf = {}

    # -- neighbour exchange: the last frame goes to the next rank, the first frame to the previous one -------------------
    # (point-to-point over xGMI: an all_gather of the boundary vertices would move world x 660 KB to every rank for the two
    # rows it needs)
    def _peer(self, r):
        """group-local rank -> the GLOBAL rank the point-to-point calls address"""
        return dist.get_global_rank(self.group, r) if self.group is not None else r

    def _gather_boundaries(self, x):
        """x (T_local, E...) -> (prev_rank_last, next_rank_first) or None at the sequence ends."""
        if self.world == 1:
            return None, None
        first, last = x[0].contiguous(), x[-1].contiguous()
        prev = None if self.is_first else torch.empty_like(last)
        nxt = None if self.is_last else torch.empty_like(first)
        ops = []
        if not self.is_first:
            p = self._peer(self.rank - 1)
            ops += [dist.P2POp(dist.isend, first, p, self.group), dist.P2POp(dist.irecv, prev, p, self.group)]
        if not self.is_last:
            p = self._peer(self.rank + 1)
            ops += [dist.P2POp(dist.isend, last, p, self.group), dist.P2POp(dist.irecv, nxt, p, self.group)]
        for req in dist.batch_isend_irecv(ops):
            req.wait()
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

    def cycle(self, row, raster=None, graphs=False, scene_update=False):
        """scene_update (single process only): launch the device-side scene update of this cycle from inside
        ``cycle_graphed``; the frame-sharded form calls ``scene_update()`` itself before the cycle"""
        e = self.e
        if self.world == 1:
            e.halo = None
            if graphs:
                e.cycle_graphed(row, raster=raster, scene_update=scene_update)
            else:
                if scene_update:
                    e.scene_device_update()
                e.cycle(row, raster=raster)
            return
        halo = {}
        pp, pn = self._gather_boundaries(e.leaf('poses_T'))
        halo['pT_prev'], halo['pT_next'] = self._static('pT_prev', pp), self._static('pT_next', pn)
        e.halo = halo                      # the velocity term runs inside cycle_begin, beside the forward
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
            dist.recv(xp, src=self._peer(self.rank - 1), group=self.group)
            dist.recv(dxp, src=self._peer(self.rank - 1), group=self.group)
            state = (xp, dxp)
        y, out = e.one_euro_shard(x, c, b, self.first_frame, state)
        if not self.is_last:
            dist.send(out[0], dst=self._peer(self.rank + 1), group=self.group)
            dist.send(out[1], dst=self._peer(self.rank + 1), group=self.group)
        return y

    def update_filters(self, c1=0.01, b1=0.02, c2=0.001, b2=0.5):
        e = self.e
        if self.world == 1 and hasattr(e, 'update_filters'):
            return e.update_filters(c1, b1, c2, b2)
        pf = self._scan(e.leaf('poses_T'), c1, b1)
        e.forward()
        vf = self._scan(e.verts.view(e.T, -1), c2, b2).view(e.verts.shape[0] // e.N, e.N, -1, 3)
        if hasattr(e, 'set_filters'):
            e.set_filters(pf, vf)            # fixed addresses: the captured cycle graphs read these buffers
        else:
            e.pT_filt, e.verts_filt = pf, vf
        a, b = self._gather_boundaries(e.verts_filt.view(e.T, -1))
        self._vf_halo = (self._static('vf_prev', a), self._static('vf_next', b))

    # -- scene aggregation across the shards (optimizer.py:578-584): the median is over ALL frames of a pixel --------
    # Pixels are sharded instead: once, every rank collects the constant inputs (normalised disparity, background
    # mask) of all frames for its slice of the pixels, pixel-major; per update the depth-range leaves of all frames
    # are all-gathered (2 floats per frame), every rank takes the median of its pixels over the whole sequence, the
    # H*W medians are all-gathered, and post-processing / un-projection / grid run redundantly on every rank (they are
    # single-image work).  Frames are padded to the longest shard with masked-out entries.
    def scene_setup(self, backmasks_local):
        e = self.e
        e.scene_device_setup(backmasks_local)
        if self.world == 1:
            return
        T, H, W = e.T, e.H, e.W
        P = H * W
        G = self.world
        tl = torch.tensor([T], device=e.dev, dtype=torch.int64)
        tls = [torch.zeros_like(tl) for _ in range(G)]
        dist.all_gather(tls, tl, group=self.group)
        Tmax = int(max(int(x.item()) for x in tls))
        Pmax = (P + G - 1) // G
        p0 = min(self.rank * Pmax, P)
        p1 = min(p0 + Pmax, P)
        d = e._scene_dev
        self._sc = dict(Tmax=Tmax, Tall=G * Tmax, Pmax=Pmax, p0=p0, p1=p1)
        assert self._sc['Tall'] <= 2048, 'pixel-sharded median: at most 2048 (padded) frames in total'
        self._sc.update(depths_t=self._gather_padded(e.depths, 0.0), back_t=self._gather_padded(d['back'], 0),
                        z=[torch.zeros(2 * Tmax, device=e.dev) for _ in range(2)], med=torch.zeros(Pmax, device=e.dev),
                        msk=torch.zeros(Pmax, device=e.dev))

    def _gather_padded(self, x, fill):
        """(T_local, P) frames of this rank -> (Pmax, G*Tmax): ALL frames (rank-major, padded with `fill`) of this
        rank's slice of the pixels, pixel-major"""
        e, sc, G = self.e, self._sc, self.world
        T, P = e.T, e.H * e.W
        Tmax, Pmax, p0, p1 = sc['Tmax'], sc['Pmax'], sc['p0'], sc['p1']
        buf = torch.full((Tmax, P), fill, dtype=x.dtype, device=e.dev)
        buf[:T] = x.view(T, P)
        outs = [torch.empty_like(buf) for _ in range(G)]
        dist.all_gather(outs, buf, group=self.group)
        full = torch.cat(outs, 0)                                  # (G*Tmax, P), rank-major frame order
        sl = torch.zeros(Pmax, G * Tmax, dtype=x.dtype, device=e.dev)
        sl[:p1 - p0] = full[:, p0:p1].t()
        return sl.contiguous()

    def _gather_pixels(self, rows):
        """(k, Pmax) per-rank pixel slices -> (k, H*W) on every rank"""
        outs = [torch.empty_like(rows) for _ in range(self.world)]
        dist.all_gather(outs, rows.contiguous(), group=self.group)
        return torch.cat(outs, 1)[:, :self.e.H * self.e.W], outs          # slices are Pmax wide, in rank order

    def scene_update(self):
        """One scene update (call at the start of a cycle >= 30, like SequenceEngine.scene_device_update)."""
        e = self.e
        if self.world == 1:
            return e.scene_device_update()
        sc, d = self._sc, e._scene_dev
        T, G = e.T, self.world
        Tmax = sc['Tmax']
        k = d['next']
        s = d['sets'][k]
        z = sc['z'][k]                # one snapshot buffer per set: the previous update's all_gather (other stream) may
        z.fill_(1.0)                  # still be reading the other one
        z[:T].copy_(e.leaf('zmin_lin').view(-1))
        z[Tmax:Tmax + T].copy_(e.leaf('zmax_lin').view(-1))
        d['ev_main'].record(e.main_stream())
        side = d['stream']
        side.wait_event(d['ev_main'])
        with e.stream_ctx(side):
            zs = [torch.empty_like(z) for _ in range(G)]
            dist.all_gather(zs, z, group=self.group)
            zmin_all = torch.cat([z[:Tmax] for z in zs]).contiguous()
            zmax_all = torch.cat([z[Tmax:] for z in zs]).contiguous()
            e.scene_median_rows(sc['depths_t'], sc['back_t'], zmin_all, zmax_all, sc['med'], sc['msk'], side)
            both = torch.stack([sc['med'], sc['msk']])
            full, outs = self._gather_pixels(both)
            d['ma_depth'].view(-1).copy_(full[0])
            d['ma_mask'].view(-1).copy_(full[1])
            e._scene_finish(s, side.cuda_stream)
            s['ev'].record(side)
            self._keep = (zs, zmin_all, zmax_all, both, outs, full)      # alive until the stream has consumed them
        d['ready'] = k
        d['next'] = 1 - k

    def scene_image(self, images_local):
        """images (T_local,H,W,3) uint8 of this rank's frames -> (scene_img (H,W,3) uint8, scene_mask (H,W)): masked
        median over ALL frames of the background colour + looped 11x11 fill (optimizer.py:595-600), once per fit.
        Frame-sharded: the same pixel sharding as the depth median (each colour plane gathered pixel-major for the own
        slice of the pixels, medians all-gathered, the fill redundantly on every rank)."""
        e = self.e
        if self.world == 1:
            return e.scene_device_image(images_local)
        sc, d = self._sc, e._scene_dev
        H, W = e.H, e.W
        main = e.main_stream()
        img = torch.as_tensor(np.ascontiguousarray(images_local)).to(e.dev)
        out = torch.empty(H, W, 3, device=e.dev)
        med, msk = torch.zeros(sc['Pmax'], device=e.dev), torch.zeros(sc['Pmax'], device=e.dev)
        m = None
        for ch in range(3):
            plane_t = self._gather_padded(img[..., ch].float().reshape(e.T, H * W), 0.0)
            e.scene_median_rows(plane_t, sc['back_t'], None, None, med, msk, main)
            full, _ = self._gather_pixels(torch.stack([med, msk]))
            val = full[0].floor().view(H, W).contiguous()             # .astype(np.uint8) of the reference
            m = full[1].view(H, W).contiguous().clone()
            e.scene_fill_plane(val, m, 11, main)
            out[..., ch] = val
        return out.clamp_(0, 255).to(torch.uint8).cpu().numpy(), m.cpu().numpy()

    def scene_swap(self):
        self.e.scene_device_swap()

    # -- logs: raw sums are all-reduced once, at the end ----------------------------------------------
    def read_log(self, rows):
        e = self.e
        e._flush_log()
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
