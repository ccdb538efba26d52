"""Frame-sharded optimisation over the GPUs of one node: one process per GPU, frames split in
contiguous blocks (block boundaries at multiples of the batch size so the in-batch terms are
unchanged), per-frame leaves and inputs local to their rank.

Per cycle the ranks exchange, over ``torch.distributed`` (backend "nccl" == RCCL on ROCm, xGMI), ONE all-reduce (sum):

  * the gradient tail of the shared leaves ``betas (N,10) | xscale (N)`` -- 11*N floats -- after which every rank applies
    the identical RMSprop step to its replica of those leaves;
  * riding in the same message (round 4), every rank's two BOUNDARY frames of the per-frame leaves ``poses_T | poses_smpl``
    (75*N floats per frame, each rank fills its own slots, the others' are zero: the sum is a gather) as they are AFTER this
    cycle's update -- the per-frame leaves need no other rank's gradients, so they are stepped before the all-reduce.  The
    next cycle's temporal terms read them: the velocity term the neighbour's translations (optimizer.py:560), the
    filtered-vertex term (:571-573) the neighbour's boundary VERTICES, which the rank skins itself from those leaves
    (``SequenceEngine._halo_forward``: one small launch pair in the side branch, bit-identical to the owner's).
    Rounds 1-3 exchanged the 83 KB x N of vertices point-to-point in the MIDDLE of the cycle, which split the cycle into
    two graph replays with the exchange on the critical path; now the cycle is one graph and nothing is communicated
    between its first kernel and its last.
  * every 25 cycles the one-euro filter state (filtered value + filtered derivative of the last local frame) is handed
    rank k -> k+1, because the filter is sequential in time (optimizer.py:664-675), and the filtered boundary vertices
    go to the neighbours once (point-to-point; outside the cycles).

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
        self._halo = None             # static halo tensors handed to the engine (world > 1)
        self._halo_ok = False         # False: the neighbours' boundary leaves must be (re-)gathered before the next cycle

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

    # -- the one message of a cycle: [ gradient tail of betas | xscale (11 N) | world x 2 boundary frames x N x (3 + 72) ] ------
    def _bufs(self):
        e = self.e
        if getattr(self, '_ar', None) is None:
            N = e.N
            dev = e.params.device
            self._nt = int(e.params.numel() - e.shared_lo)
            self._slot = 2 * N * 75
            self._ar = torch.zeros(self._nt + self.world * self._slot, dtype=torch.float32, device=dev)
            nh = (0 if self.is_first else N) + (0 if self.is_last else N)
            self._halo = dict(has_prev=not self.is_first, has_next=not self.is_last,
                              poses=torch.zeros(nh, 72, dtype=torch.float32, device=dev) if nh else None,
                              transl=torch.zeros(nh, 3, dtype=torch.float32, device=dev) if nh else None,
                              pT_prev=None, pT_next=None, vf_prev=None, vf_next=None)
            k = 0
            if not self.is_first:
                self._halo['pT_prev'] = self._halo['transl'][:N]
                k = N
            if not self.is_last:
                self._halo['pT_next'] = self._halo['transl'][k:k + N]
        return self._ar

    def _slot_of(self, r):
        N = self.e.N
        o = self._nt + r * self._slot
        return self._ar[o:o + self._slot].view(2, N, 75)          # [first frame | last frame][person][pT(3) | pose(72)]

    def _pack(self, with_grads=True):
        """own boundary leaves (and the shared gradient tail) into the message; everybody else's slots zero"""
        e, ar = self.e, self._ar
        ar.zero_()
        if with_grads:
            ar[:self._nt].copy_(e.grads[e.shared_lo:])
        pT, ps = e.leaf('poses_T'), e.leaf('poses_smpl')
        own = self._slot_of(self.rank)
        own[0, :, :3].copy_(pT[0]); own[0, :, 3:].copy_(ps[0])
        own[1, :, :3].copy_(pT[-1]); own[1, :, 3:].copy_(ps[-1])

    def _unpack(self, with_grads=True):
        e, h, N = self.e, self._halo, self.e.N
        if with_grads:
            e.grads[e.shared_lo:].copy_(self._ar[:self._nt])
        k = 0
        if not self.is_first:
            prev = self._slot_of(self.rank - 1)[1]                # the previous rank's LAST frame
            h['transl'][:N].copy_(prev[:, :3]); h['poses'][:N].copy_(prev[:, 3:])
            k = N
        if not self.is_last:
            nxt = self._slot_of(self.rank + 1)[0]                 # the next rank's FIRST frame
            h['transl'][k:k + N].copy_(nxt[:, :3]); h['poses'][k:k + N].copy_(nxt[:, 3:])

    def _run(self, key, fn, graphs):
        """a fixed sequence of small copies on static buffers: through the engine's captured-graph facility when it has one"""
        if graphs and hasattr(self.e, 'replay'):
            self.e.replay(key, fn, wait_scene=False)
        else:
            fn()

    def refresh_halo(self):
        """COLLECTIVE: gather the neighbours' boundary leaves as they are now (first cycle of a fit; after the leaves were
        set from outside).  Inside a fit every cycle's all-reduce carries them."""
        if self.world == 1:
            return
        self._bufs()
        self._pack(with_grads=False)
        dist.all_reduce(self._ar, op=dist.ReduceOp.SUM, group=self.group)
        self._unpack(with_grads=False)
        self._halo_ok = True

    def leaves_changed(self):
        """the per-frame leaves were written from outside a cycle (set_leaves, the start of a fit, a test): the neighbours'
        copies of the boundary frames are stale -- every rank must call ``refresh_halo()`` (a COLLECTIVE) before its next
        ``cycle`` -- and the device-style learning-rate schedule of ``step(lr=None)`` starts over, as a new optimiser's would"""
        self._halo_ok = False
        self._lr32 = None

    def cycle(self, row, raster=None, graphs=False, scene_update=False):
        """scene_update (single process only): launch the device-side scene update of this cycle from inside
        ``cycle_graphed``; the frame-sharded form calls ``scene_update()`` itself before the cycle.
        Frame-sharded: the cycle itself communicates nothing -- ``step`` does (one all-reduce)."""
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
        self._bufs()
        if not self._halo_ok:
            # no hidden collective: a cycle issued on some ranks only must not deadlock the others inside an all-reduce
            raise RuntimeError('frame-sharded cycle with stale halos: the per-frame leaves were set from outside a cycle -- call '
                               'refresh_halo() on EVERY rank first (a collective; fit() and refresh_global_leaves() do)')
        h = self._halo
        h['vf_prev'], h['vf_next'] = self._vf_halo if self._vf_halo is not None else (None, None)
        e.halo = h
        self._graphs = bool(graphs)
        if graphs and hasattr(e, 'cycle_graphed'):
            e.cycle_graphed(row, raster=raster)                   # ONE graph replay; the log row travels with the next step
        else:
            e.cycle(row, raster=raster)

    def step(self, lr=None):
        """lr given: host-side schedule; lr None: the device-resident schedule (single process only).
        Frame-sharded: per-frame leaves first (local gradients only), then THE all-reduce of the cycle -- shared gradient
        tail + everybody's updated boundary leaves -- then the shared leaves."""
        e = self.e
        if self.world == 1:
            if lr is None:
                e.step_dev()
            else:
                e.step(lr)
            return
        if lr is None:
            # the device-resident schedule of step_dev (lr0 = 0.01, x 0.99 per step, in float32), kept on the host here
            if getattr(self, '_lr32', None) is None:
                self._lr32 = np.float32(0.01)
            lr = float(self._lr32)
            self._lr32 = np.float32(self._lr32 * np.float32(0.99))
        g = getattr(self, '_graphs', False)
        e.step_local(lr)
        if g and self._ar_in_graph():
            # MHHIP_SHARD_AR_GRAPH=1: pack -> all-reduce -> unpack as ONE captured graph (RCCL collectives capture and replay:
            # tests/test_rccl_single_rank_gpu.py on the one-rank group; off by default until an N-rank node has run it)
            def message():
                self._pack()
                dist.all_reduce(self._ar, op=dist.ReduceOp.SUM, group=self.group)
                self._unpack()
            self._run(('shard_message',), message, True)
        else:
            self._run(('shard_pack',), self._pack, g)
            dist.all_reduce(self._ar, op=dist.ReduceOp.SUM, group=self.group)
            self._run(('shard_unpack',), self._unpack, g)
        e.step_shared(lr)
        self._halo_ok = True

    def _ar_in_graph(self):
        if getattr(self, '_ar_graph', None) is None:
            import os
            on = os.environ.get('MHHIP_SHARD_AR_GRAPH') == '1' and hasattr(self.e, 'replay')
            try:
                on = on and str(dist.get_backend(self.group)) == 'nccl'       # host-staged dry runs cannot be captured
            except Exception:
                on = False
            self._ar_graph = bool(on)
        return self._ar_graph

    # -- one-euro filters with the state handed down the ranks (optimizer.py:383-392) ----------------
    def _scan(self, x, c, b, chunks=None):
        """one-euro filter of this rank's frames of x (T_local, ...), the recurrence's state handed rank k -> k + 1.
        The hand-off is a serial chain of world - 1 hops -- so the elements go in CHUNKS (round 6): rank k scans chunk j as soon
        as its state has arrived and sends the result on before it touches chunk j + 1, i.e. rank k + 1 works on chunk j
        while rank k works on chunk j + 1: the chain costs (world - 1 + chunks - 1) chunk steps instead of (world - 1) x chunks.
        Elements are independent of each other: the same bits as the one-piece form (tests/test_sharded_world_cpu.py).
        MHHIP_SHARD_SCAN_CHUNKS (default 4; 1 = the one-piece hand-off of rounds 3-5)."""
        import os
        e = self.e
        T = x.shape[0]
        x2 = x.reshape(T, -1)
        E = x2.shape[1]
        if chunks is None:
            chunks = int(os.environ.get('MHHIP_SHARD_SCAN_CHUNKS', '4'))
        chunks = max(1, min(int(chunks), E))
        if self.world == 1 or chunks == 1:
            state = None
            if not self.is_first:
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
        bounds = [(j * E) // chunks for j in range(chunks + 1)]
        y = torch.empty_like(x2)
        for j in range(chunks):
            lo, hi = bounds[j], bounds[j + 1]
            state = None
            if not self.is_first:
                st2 = torch.empty(2, hi - lo, dtype=torch.float32, device=x.device)
                dist.recv(st2, src=self._peer(self.rank - 1), group=self.group)
                state = (st2[0].contiguous(), st2[1].contiguous())
            yj, out = e.one_euro_shard(x2[:, lo:hi].contiguous(), c, b, self.first_frame, state)
            y[:, lo:hi] = yj.reshape(T, hi - lo)
            if not self.is_last:
                # (plain send / recv: stream-ordered on nccl, host-blocking on gloo -- either way rank k + 1 has chunk j while
                # this rank goes on to chunk j + 1)
                dist.send(torch.stack([out[0].reshape(-1), out[1].reshape(-1)]).contiguous(), dst=self._peer(self.rank + 1), group=self.group)
        return y.view(x.shape)

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
