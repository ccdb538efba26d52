"""torch-CPU stand-in for ``mhhip.sequence.SequenceEngine`` with the same duck-typed interface, so
that the frame-sharded driver (mhhip/sharded.py: halo exchange, shared-gradient all-reduce, one-euro
state hand-off) can be exercised with world_size > 1 on the gloo backend without a GPU.  The maths
comes from the oracle (tests may use it); only the raster-free terms are modelled."""
import contextlib
import math

import numpy as np
import torch

from oracle import fit_oracle as fo
from oracle import lbs_oracle as lo
from oracle import scene_oracle as so


class _NullStream(object):
    """what the sharded driver needs of a stream on a host without one"""
    cuda_stream = None

    def wait_event(self, ev):
        pass

    def synchronize(self):
        pass


class _NullEvent(object):
    def record(self, stream=None):
        pass


def one_euro_shard_np(x, min_cutoff, beta, first_frame, state, frame_rate=25):
    y = np.array(x, dtype=np.float32, copy=True)
    two_pi = 2 * math.pi
    if state is None:
        x_prev, dx_prev = y[0].copy(), np.zeros_like(y[0])
        t_prev = np.zeros_like(y[0])
        start = 1
    else:
        x_prev, dx_prev = state[0].reshape(y[0].shape).copy(), state[1].reshape(y[0].shape).copy()
        t = np.float32(0)
        for i in range(1, first_frame):
            t = np.float32(t + np.float32(i / frame_rate))
        t_prev = np.full_like(y[0], t)
        start = 0
    time_i = t_prev.copy()
    for k in range(start, len(y)):
        i = first_frame + k
        time_i = time_i + (i / frame_rate)
        xi = y[k].copy()
        t_e = time_i - t_prev
        r = two_pi * 1.0 * t_e
        a_d = r / (r + 1)
        dx_hat = a_d * ((xi - x_prev) / t_e) + (1 - a_d) * dx_prev
        r = two_pi * (float(min_cutoff) + float(beta) * np.abs(dx_hat)) * t_e
        a = r / (r + 1)
        x_hat = a * xi + (1 - a) * x_prev
        x_prev, dx_prev, t_prev = x_hat, dx_hat, time_i
        y[k] = x_hat
    return y, (x_prev.reshape(-1).copy(), dx_prev.reshape(-1).copy())


class CpuShardEngine(object):
    def __init__(self, model, image_size, T, N, cam_K, coefs, batch_size, pose2d, poses_ref, valid, betas_ref):
        self.model, self.T, self.N, self.B = model, T, N, T * N
        self.W, self.H = image_size
        self.K = torch.tensor(np.asarray(cam_K, np.float32))[None]
        self.c = coefs
        self.batch = batch_size
        self.nbatches = (T + batch_size - 1) // batch_size
        B = self.B
        self.sizes = [B * 3, B * 72, T, T, N * 10, N]
        self.offs = np.concatenate([[0], np.cumsum(self.sizes)]).astype(np.int64)
        n = int(self.offs[-1])
        self.params, self.grads = torch.zeros(n), torch.zeros(n)
        self.sq, self.buf = torch.zeros(n), torch.zeros(n)
        self.shared_lo = int(self.offs[4])
        self.pose2d = torch.tensor(pose2d).view(B, 17, 3)
        self.poses_ref = torch.tensor(poses_ref).view(B, 72)
        self.valid = torch.tensor(valid).view(B, 1)
        self.betas_ref = torch.tensor(betas_ref).view(N, 10)
        self.verts = None
        self.verts_filt = None
        self.pT_filt = None
        self.halo = None
        self.log = torch.zeros(64, 16)
        self.dev = torch.device('cpu')

    # -- the part of SequenceEngine's interface the drop-in optimiser drives (tests/test_fit_sharded_cpu.py) --------
    @classmethod
    def factory(cls, model):
        def make(image_size, num_frames, num_people, cam_K, cam_dist_coef=None, coefs=None, joint_confidence_thr=0.5,
                 eps=1e-3, batch_size=10, joint_weights=None):
            T, N = num_frames, num_people
            z = lambda *s: np.zeros(s, np.float32)
            e = cls(model, image_size, T, N, cam_K, coefs, batch_size, z(T, N, 17, 3), z(T, N, 72), z(T, N, 1), z(N, 10))
            e.dev = torch.device('cpu')
            e.has_images = False
            e._scene_dev = None
            return e
        return make

    def set_leaves(self, poses_T, poses_smpl, betas, zmin_lin, zmax_lin, xscale=None):
        f = lambda a: torch.tensor(np.asarray(a, np.float32))
        self.leaf('poses_T').copy_(f(poses_T).view(self.T, self.N, 3))
        self.leaf('poses_smpl').copy_(f(poses_smpl).view(self.T, self.N, 72))
        self.leaf('betas').copy_(f(betas).view(self.N, 10))
        self.leaf('zmin_lin').copy_(f(zmin_lin).view(self.T))
        self.leaf('zmax_lin').copy_(f(zmax_lin).view(self.T))
        self.leaf('xscale').copy_(f(xscale).view(self.N) if xscale is not None else torch.zeros(self.N))
        self.sq.zero_()
        self.buf.zero_()

    def set_batch_size(self, batch_size):
        self.batch = int(batch_size)
        self.nbatches = (self.T + self.batch - 1) // self.batch

    def set_batch_table(self, table):
        assert table is None, 'the stand-in models the raster-free terms: no foot sliding, no batch table'

    def stage(self, pose2d, poses_ref, valid, betas_ref, seg_mask=None, depths=None):
        B, N = self.B, self.N
        self.pose2d = torch.tensor(np.asarray(pose2d, np.float32)).view(B, 17, 3)
        self.poses_ref = torch.tensor(np.asarray(poses_ref, np.float32)).view(B, 72)
        self.valid = torch.tensor(np.asarray(valid, np.float32)).view(B, 1)
        self.betas_ref = torch.tensor(np.asarray(betas_ref, np.float32)).view(N, 10)

    def update_filters(self, c1=0.01, b1=0.02, c2=0.001, b2=0.5):
        pf, _ = self.one_euro_shard(self.leaf('poses_T'), c1, b1, 0)
        self.forward()
        vf, _ = self.one_euro_shard(self.verts.view(self.T, -1), c2, b2, 0)
        self.pT_filt, self.verts_filt = pf, vf.view(self.T, self.N, -1, 3)

    def leaf(self, name, buf=None):
        buf = self.params if buf is None else buf
        T, N = self.T, self.N
        i, shape = {'poses_T': (0, (T, N, 3)), 'poses_smpl': (1, (T, N, 72)), 'zmin_lin': (2, (T,)),
                    'zmax_lin': (3, (T,)), 'betas': (4, (N, 10)), 'xscale': (5, (N,))}[name]
        return buf[int(self.offs[i]):int(self.offs[i + 1])].view(*shape)

    def _verts(self, p):
        T, N = self.T, self.N
        be = self.leaf('betas', p).unsqueeze(0).expand(T, N, 10).reshape(-1, 10)
        out = lo.smpl_forward(self.model, be, self.leaf('poses_smpl', p).reshape(-1, 72))
        s = torch.pow(torch.tensor(1.1), self.leaf('xscale', p)).view(1, N, 1, 1)
        pT = self.leaf('poses_T', p).view(T, N, 1, 3)
        return s * out['verts'].view(T, N, -1, 3) + pT, s * out['joints_alphapose'].view(T, N, 17, 3) + pT

    def forward(self):
        with torch.no_grad():
            self.verts = self._verts(self.params)[0].reshape(self.B, -1, 3).clone()

    def cycle(self, row, use_images=True, raster=None):
        self.cycle_begin()
        self.cycle_finish(row, use_images, raster)

    def cycle_begin(self):
        self.grads.zero_()
        self.forward()

    def cycle_finish(self, row, use_images=True, raster=None):
        T, N, c = self.T, self.N, self.c
        p = self.params.clone().requires_grad_(True)
        verts, joints = self._verts(p)
        uv = fo.project_points(joints.view(-1, 17, 3), self.K.expand(self.B, 3, 3)).view(self.B, 17, 2)
        conf = (self.pose2d[..., 2:3] >= 0.5).float()
        nrm = torch.tensor([float(self.W), float(self.H)])
        l2d = ((conf * uv / nrm - conf * self.pose2d[..., :2] / nrm) ** 2).sum()
        pr = (self.valid * self.poses_ref - self.valid * self.leaf('poses_smpl', p).view(-1, 72)).abs().sum()
        lb = T * (self.leaf('betas', p) - self.betas_ref).abs().sum()
        s = torch.pow(torch.tensor(1.1), self.leaf('xscale', p))
        s_avg, s_per = ((s - 1).sum()) ** 2, ((s - 1) ** 2).mean()
        h = dict(self.halo or {})
        if h.get('poses') is not None:
            # the neighbours' boundary frames skinned here from their leaves (SequenceEngine._halo_forward)
            with torch.no_grad():
                nh = h['poses'].shape[0]
                be = self.leaf('betas').repeat(nh // N, 1)
                hv = lo.smpl_forward(self.model, be, h['poses'])['verts'].view(nh // N, N, -1, 3)
                hv = torch.pow(torch.tensor(1.1), self.leaf('xscale')).view(1, N, 1, 1) * hv + h['transl'].view(nh // N, N, 1, 3)
            k = 0
            if h.get('has_prev'):
                h['v_prev'] = hv[0]
                k = 1
            if h.get('has_next'):
                h['v_next'] = hv[k]
        pT = self.leaf('poses_T', p)
        seq = [pT] if h.get('pT_prev') is None else [h['pT_prev'].view(1, N, 3), pT]
        own = torch.cat(seq)                                   # pairs (t-1, t) owned by the owner of t
        vel = ((own[1:] - own[:-1]) ** 2).sum()
        if h.get('pT_next') is not None:                       # the neighbour's pair still pulls on our last frame
            vel_n = ((h['pT_next'].view(N, 3) - pT[-1]) ** 2).sum()
        else:
            vel_n = torch.zeros(())
        filt = torch.zeros(())
        filt_n = torch.zeros(())
        if self.verts_filt is not None and self.pT_filt is not None:
            v = verts.view(T, -1)
            vf = self.verts_filt.view(T, -1)
            if h.get('v_prev') is not None:
                v = torch.cat([h['v_prev'].view(1, -1), v])
                vf = torch.cat([h['vf_prev'].view(1, -1), vf])
            filt = (((v[1:] - v[:-1]) - (vf[1:] - vf[:-1])) ** 2).sum()
            if h.get('v_next') is not None:
                filt_n = (((h['v_next'].view(-1) - v[-1]) - (h['vf_next'].view(-1) - vf[-1])) ** 2).sum()
        total = (c['proj2d'] * l2d + c['reg_poses'] * (pr + lb)
                 + self.nbatches * (c['reg_scales'] * s_per + float(c['reg_scales'] > 0) * s_avg)
                 + c['reg_velocity'] * (vel + vel_n) + c['reg_verts_filter'] * (filt + filt_n))
        total.backward()
        self.grads.copy_(p.grad)
        row_t = torch.zeros(16)
        row_t[0], row_t[3], row_t[9], row_t[10], row_t[11] = l2d.detach(), pr.detach(), lb.detach(), s_avg.detach(), s_per.detach()
        row_t[7], row_t[8] = vel.detach(), filt.detach()
        self.log[row] = row_t

    def step(self, lr):
        fo.rmsprop_step(self.params, self.grads, self.sq, self.buf, lr)

    def step_local(self, lr):
        lo_ = self.shared_lo
        fo.rmsprop_step(self.params[:lo_], self.grads[:lo_], self.sq[:lo_], self.buf[:lo_], lr)

    def step_shared(self, lr):
        lo_ = self.shared_lo
        fo.rmsprop_step(self.params[lo_:], self.grads[lo_:], self.sq[lo_:], self.buf[lo_:], lr)

    # -- scene aggregation hooks (same contract as SequenceEngine; arithmetic from oracle/scene_oracle.py) ------------
    def main_stream(self):
        return _NullStream()

    def stream_ctx(self, stream):
        return contextlib.nullcontext()

    def set_images(self, depths):
        self.depths = torch.tensor(np.asarray(depths, np.float32))
        self.has_images = True

    def scene_device_setup(self, backmasks):
        H, W, T = self.H, self.W, self.T
        z = lambda *s: torch.zeros(*s)
        self._scene_dev = dict(back=torch.tensor((np.asarray(backmasks) != 0).astype(np.uint8)), ws=None, ma_depth=z(H, W),
                               ma_mask=z(H, W), depth=z(H, W), stream=_NullStream(), ev_main=_NullEvent(),
                               sets=[dict(pts=z(H * W, 3), count=torch.zeros(1, dtype=torch.int32), grid=None,
                                          zsnap=z(2 * T), ev=_NullEvent()) for _ in range(2)],
                               next=0, ready=None, front=None)

    def scene_median_rows(self, depths_t, back_t, zmin, zmax, med, msk, stream):
        d = depths_t.numpy().astype(np.float32)
        if zmin is not None:
            zmin, zmax = zmin.numpy().astype(np.float32), zmax.numpy().astype(np.float32)
            min_z = np.log(np.float32(1) + np.exp(zmin)).astype(np.float32)
            max_z = (min_z + np.float32(1) + np.log(np.float32(1) + np.exp(zmax))).astype(np.float32)
            inv_min, inv_max = (np.float32(1) / min_z)[None], (np.float32(1) / max_z)[None]
            d = (np.float32(1) / (d * (inv_min - inv_max) + inv_max)).astype(np.float32)
        m = np.ma.median(np.ma.array(d, mask=back_t.numpy() == 0), axis=1)
        seen = back_t.numpy().max(axis=1) > 0
        med.copy_(torch.tensor(np.where(seen, np.ma.filled(m, 0.0), 0.0).astype(np.float32)))
        msk.copy_(torch.tensor(seen.astype(np.float32)))

    def _scene_finish(self, s, st):
        d = self._scene_dev
        ma_depth, ma_mask = d['ma_depth'].numpy(), d['ma_mask'].numpy()
        with np.errstate(invalid='ignore', divide='ignore'):
            depth = so.postprocess_depthmap(ma_depth, ma_mask, use_bilateral_filter=True)
        d['depth'].copy_(torch.tensor(depth))
        H, W = self.H, self.W
        K = self.K[0].numpy()
        u = (np.arange(W, dtype=np.float32) + 0.5 - K[0, 2]) / K[0, 0]
        v = (np.arange(H, dtype=np.float32) + 0.5 - K[1, 2]) / K[1, 1]
        pts = np.stack([depth * u[None, :], depth * v[:, None], depth], -1)[ma_mask > 0.5].astype(np.float32)
        s['pts'][:len(pts)] = torch.tensor(pts)
        s['count'][0] = len(pts)

    def scene_device_update(self):
        d = self._scene_dev
        T, P = self.T, self.H * self.W
        k = d['next']
        med, msk = torch.zeros(P), torch.zeros(P)
        self.scene_median_rows(self.depths.view(T, P).t().contiguous(), d['back'].view(T, P).t().contiguous(),
                               self.leaf('zmin_lin').clone(), self.leaf('zmax_lin').clone(), med, msk, None)
        d['ma_depth'].view(-1).copy_(med)
        d['ma_mask'].view(-1).copy_(msk)
        self._scene_finish(d['sets'][k], None)
        d['ready'], d['next'] = k, 1 - k

    def scene_device_swap(self):
        d = self._scene_dev
        if d['ready'] is not None:
            d['front'], d['ready'] = d['sets'][d['ready']], None

    def scene_device_result(self):
        d = self._scene_dev
        s = d['front'] if d['ready'] is None else d['sets'][d['ready']]
        n = int(s['count'].item())
        return d['depth'].numpy().copy(), d['ma_mask'].numpy() > 0.5, s['pts'][:n].clone()

    def scene_fill_plane(self, val, mask, ksize, stream):
        x, m = val.numpy().astype(np.uint8), mask.numpy().copy()      # integer plane: every sweep stores floor(median)
        while m.min() == 0:
            x, m = so.fillin_values(x, m, filter_size=ksize)
        val.copy_(torch.tensor(x.astype(np.float32)))
        mask.copy_(torch.tensor(m))

    def scene_device_image(self, images):
        T, H, W = self.T, self.H, self.W
        back = self._scene_dev['back'].numpy()
        img, _, _ = so.aggregate_scene_median(None, np.asarray(images), back, images_only=True)
        mask = (back.max(axis=0) > 0).astype(np.float32)
        while mask.min() == 0:
            img, mask = so.fillin_values(img, mask, filter_size=11)
        return img, mask

    def one_euro_shard(self, x, min_cutoff, beta, first_frame, state_in=None):
        st = None if state_in is None else (state_in[0].numpy(), state_in[1].numpy())
        y, out = one_euro_shard_np(x.detach().numpy(), min_cutoff, beta, first_frame, st)
        return torch.tensor(y), (torch.tensor(out[0]), torch.tensor(out[1]))

    def _flush_log(self):
        """(the device engine parks a replayed cycle's log row until its update's launch; rows are written at once here)"""

    def read_log(self, rows, nbatches_total=None):
        raw = self.log[:rows].numpy().astype(np.float64)
        nb = float(nbatches_total or self.nbatches)
        return [{'loss_pose24j': r[0] / nb, 'reg_ref_poses': (r[3] + r[9]) / nb, 'reg_scale': r[10] + r[11],
                 'reg_vel': r[7], 'reg_filter_verts': r[8]} for r in raw]


def cpu_optimizer_class(model):
    """the drop-in optimiser with the torch-CPU stand-in above as its engine: what the gloo tests of the frame-sharded
    orchestration construct (the product class refuses a non-HIP device and always builds a ``SequenceEngine``)"""
    from mhmocap.optimizer import SMPLDepthSequenceOptimizer

    class CpuShardOptimizer(SMPLDepthSequenceOptimizer):
        _needs_hip = False

        def _make_engine(self, **kw):
            return CpuShardEngine.factory(model)(**kw)
    return CpuShardOptimizer
