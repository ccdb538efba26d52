"""ORACLE (test infrastructure only) -- CPU restatement of the sequence optimiser.

Restates ``mhmocap/optimizer.py`` (+ the two used loss builders of
``mhmocap/losses.py``, the camera helpers of ``mhmocap/transforms.py``, the
binary erosion of ``mhmocap/morphology.py`` and ``mhmocap/one_euro_filter.py``)
in the build's own words on torch-CPU tensors; gradients come from autograd.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  Pinned against reference-generated fixtures by
``tests/test_oracle_golden.py``.  The rasteriser-fed terms call
``oracle/raster_oracle.py`` (PyTorch3D itself is not available anywhere in this
project: those two terms are "parity unpinned").
"""
import math

import numpy as np
import torch

from . import lbs_oracle


# ------------------------------------------------------------------------------------------
# camera helpers (transforms.py)
# ------------------------------------------------------------------------------------------

def focal_from_fov(side, fov_deg):
    """transforms.py:263-265"""
    return 0.5 * side / math.tan(math.pi * fov_deg / 360.0)


def project_points(pts, K, Kd=None):
    """pts (B,M,3), K (B,3,3) -> pixels (B,M,2); transforms.py:74-95."""
    xy = pts[..., :2] / pts[..., 2:3]
    if Kd is not None:                                             # :78-90
        x, y = xy[..., 0], xy[..., 1]
        r = x * x + y * y
        rad = 1 + Kd[0] * r + Kd[1] * r * r + Kd[4] * r * r * r
        xx = x * rad + 2 * Kd[2] * x * y + Kd[3] * (r + 2 * x * x)
        yy = y * rad + 2 * Kd[3] * y * y + Kd[2] * (r + 2 * y * y)     # (sic) :85-88
        xy = torch.stack([xx, yy], dim=-1)
    Kt = K.transpose(1, 2)
    return torch.bmm(xy, Kt[:, :2, :2]) + Kt[:, 2:, :2]            # :92


def unproject_points(uvd, K):
    """uvd (B,M,3), K (B,3,3) -> xyz (B,M,3); transforms.py:114-130."""
    Kt = K.transpose(1, 2)
    xy = uvd[..., 2:3] * ((uvd[..., :2] - Kt[:, 2:3, 0:2]) @ torch.linalg.inv(Kt[:, :2, :2]))
    return torch.cat([xy, uvd[..., 2:3]], dim=-1)


def calibration_matrix_ndc(znear, zfar, cam_K, image_size):
    """4x4 NDC projection handed to the PyTorch3D camera; transforms.py:222-255.
    ``image_size`` is (W, H)."""
    W, H = image_size
    if W > H:
        s = 2 * cam_K[1, 1] / H
        u = W / H
        w1 = u * (W - 2 * cam_K[0, 2]) / W
        h1 = (H - 2 * cam_K[1, 2]) / H
    elif H > W:
        s = 2 * cam_K[0, 0] / W
        u = H / W
        w1 = (W - 2 * cam_K[0, 2]) / W
        h1 = u * (H - 2 * cam_K[1, 2]) / H
    else:
        s = 2 * (cam_K[0, 0] + cam_K[1, 1]) / (W + H)
        w1 = (W - 2 * cam_K[0, 2]) / W
        h1 = (H - 2 * cam_K[1, 2]) / H
    f1 = zfar / (zfar - znear)
    f2 = -(zfar * znear) / (zfar - znear)
    return np.array([[s, 0, w1, 0], [0, s, h1, 0], [0, 0, f1, f2], [0, 0, 1, 0]], np.float32)


def softplus(x):
    """transforms.py:296-297 (naive form, kept)."""
    return torch.log(1.0 + torch.exp(x))


# ------------------------------------------------------------------------------------------
# losses (losses.py), erosion (morphology.py), one-euro (one_euro_filter.py)
# ------------------------------------------------------------------------------------------

def avg_log_depth_loss(pred_disp, true_disp, mask, eps=1e-3):
    """losses.py:19-30: per (frame, person) squared difference of masked mean log-disparity."""
    lp = mask * torch.log(torch.clamp(pred_disp, min=eps))
    lt = mask * torch.log(torch.clamp(true_disp, min=eps))
    cnt = mask.sum(dim=(2, 3)) + 1
    return ((lp.sum(dim=(2, 3)) / cnt - lt.sum(dim=(2, 3)) / cnt) ** 2).sum()


def masked_mse_loss(a, b, mask):
    """losses.py:33-40"""
    return ((mask * (a - b)) ** 2).sum() / (mask.sum() + 1.0)


def erode3x3(x):
    """One ``Erode2D(3)`` (morphology.py:29-31) on (...,H,W) {0,1} maps: a pixel survives when
    no pixel of its 3x3 neighbourhood is < 0.5; pixels outside the image do not count
    (zero padding of the *inverted* map)."""
    bad = (x < 0.5).to(x.dtype)
    H, W = x.shape[-2:]
    pad = torch.nn.functional.pad(bad, (1, 1, 1, 1))
    acc = torch.zeros_like(bad)
    for dy in range(3):
        for dx in range(3):
            acc = acc + pad[..., dy:dy + H, dx:dx + W]
    return 1 - torch.clamp(acc, 0, 1)


def one_euro_sequence(x, min_cutoff, beta, frame_rate=25):
    """optimizer.py:664-675 driving one_euro_filter.py:32-53 (d_cutoff=1).

    numpy float32 semantics are reproduced: the time stamp is a float32 array that
    accumulates ``i/frame_rate`` (so ``t_e`` is the float32 difference of two running sums),
    python scalars are weak (NEP 50).
    """
    y = np.array(x, dtype=np.float32, copy=True)
    t_prev = np.zeros_like(y[0])
    x_prev = y[0].copy()
    dx_prev = np.zeros_like(y[0])
    time_i = np.zeros_like(y[0])
    two_pi = 2 * math.pi
    for i in range(1, len(y)):
        time_i = time_i + (i / frame_rate)
        xi = y[i].copy()
        t_e = time_i - t_prev
        r = two_pi * 1.0 * t_e
        a_d = r / (r + 1)
        dx = (xi - x_prev) / t_e
        dx_hat = a_d * dx + (1 - a_d) * dx_prev
        cutoff = float(min_cutoff) + float(beta) * np.abs(dx_hat)
        r = two_pi * cutoff * t_e
        a = r / (r + 1)
        x_hat = a * xi + (1 - a) * x_prev
        x_prev, dx_prev, t_prev = x_hat, dx_hat, time_i
        y[i] = x_hat
    return y


# ------------------------------------------------------------------------------------------
# optimiser updates (torch.optim semantics restated; optimizer.py:355-356, 738-739)
# ------------------------------------------------------------------------------------------

def rmsprop_step(p, g, sq, buf, lr, alpha=0.5, momentum=0.9, eps=1e-8):
    """In place: torch.optim.RMSprop(lr, alpha, momentum), not centred."""
    sq.mul_(alpha).addcmul_(g, g, value=1 - alpha)
    avg = sq.sqrt().add_(eps)
    buf.mul_(momentum).addcdiv_(g, avg)
    p.add_(buf, alpha=-lr)


def adam_step(p, g, m, v, step, lr, b1=0.5, b2=0.5, eps=1e-6):
    """In place: torch.optim.Adam (no amsgrad, no weight decay); ``step`` counts from 1."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


# ------------------------------------------------------------------------------------------
# the sequence optimiser
# ------------------------------------------------------------------------------------------

class SequenceOracle(object):
    """Same state and maths as ``SMPLDepthSequenceOptimizer`` (optimizer.py:146-770)."""

    def __init__(self, model, image_size, num_frames, cam_K, cam_dist_coef=None, coefs=None,
                 joint_confidence_thr=0.5, eps=1e-3, znear=1.0, zfar=100.0, rasteriser=None):
        self.model = model
        self.W, self.H = image_size
        self.T = num_frames
        self.cam_K = np.asarray(cam_K, np.float32)
        self.Kd = cam_dist_coef
        c = dict(proj2d=1.0, depth=1.0, silhouette=1.0, reg_velocity=1.0, reg_verts_filter=1.0,
                 reg_poses=1.0, reg_scales=1.0, reg_contact=1.0, reg_foot_sliding=1.0)
        c.update(coefs or {})
        self.c = c
        self.thr = joint_confidence_thr
        self.eps = eps
        self.rasteriser = rasteriser        # callable(verts (B,V,3)) -> zbuf (B,H,W), alpha (B,H,W)
        self.Kt = torch.tensor(self.cam_K)[None]
        self.scene_pcd = None
        self.scene_depth = None
        self.pT_filt = None
        self.v_filt = None
        self.joint_w = torch.ones(1, 1, 17, 1)          # optimizer.py:128-130, 259 (uniform -> 1)

    # -- optimizer.py:710-770 ---------------------------------------------------------------
    def init_global_poses(self, pose2d, poses_smpl, betas_smpl, num_iter, joints_thr=0.15):
        T, N = pose2d.shape[:2]
        pT = torch.tensor(np.tile(np.array([[[[0, 0, 1]]]], np.float32), (T, N, 1, 1)), requires_grad=True)
        th = torch.tensor(poses_smpl.astype(np.float32)).view(T * N, 72)
        be = torch.tensor(betas_smpl.astype(np.float32)).view(T * N, 10)
        vis = torch.tensor((pose2d[..., 2:] > joints_thr).astype(np.float32))
        gt = torch.tensor(pose2d[..., :2].astype(np.float32))
        with torch.no_grad():
            j17 = lbs_oracle.smpl_forward(self.model, be, th)[getattr(self, 'joints_key', 'joints_alphapose')].view(T, N, 17, 3)
        m = torch.zeros_like(pT)
        v = torch.zeros_like(pT)
        lr = 0.5
        log = []
        scale = torch.pow(torch.tensor(1.1), self.xscale.detach())
        for it in range(num_iter):
            if pT.grad is not None:
                pT.grad = None
            g3 = scale * j17 + pT
            uv = project_points(g3.view(T * N, 17, 3), self.Kt.expand(T * N, 3, 3), self.Kd).view(T, N, 17, 2)
            w = self.joint_w * vis
            l2d = torch.mean((w * uv - w * gt) ** 2)                               # :754-756 (mean)
            speed = ((pT[1:] - pT[:-1]) ** 2).sum()                                # :758
            loss = self.c['proj2d'] * l2d + self.c['reg_velocity'] * speed
            log.append(float(l2d.detach()))
            loss.backward()
            with torch.no_grad():
                adam_step(pT, pT.grad, m, v, it + 1, lr)
            lr *= 0.95
        return pT.detach().numpy().copy(), log

    # -- optimizer.py:262-321 ---------------------------------------------------------------
    def init_optimized_variables(self, pose2d, poses_smpl, betas_smpl, valid_smpl, num_iter=100,
                                 poses_T=None, scale_factor=None):
        T, N = pose2d.shape[:2]
        self.N = N
        if scale_factor is not None:      # optimizer.py:277-280: a given person scale is a constant (1.1 ** x = scale), not a leaf
            x = (np.log(np.asarray(scale_factor)) / np.log(1.1)).astype(np.float32)      # in the caller's dtype, as :280 does
            self.xscale = torch.tensor(x).view(1, N, 1, 1)
        else:
            self.xscale = torch.zeros(1, N, 1, 1, requires_grad=True)
        log = []
        if poses_T is None:
            poses_T, log = self.init_global_poses(pose2d, poses_smpl, betas_smpl, num_iter)
        self.poses_T = torch.tensor(poses_T, requires_grad=True)
        max_z = np.clip(np.max(poses_T[..., 2:], axis=1), 2, None)                # :292
        self.poses_smpl = torch.tensor(poses_smpl.astype(np.float32), requires_grad=True)
        avg = np.mean(betas_smpl, axis=0, keepdims=True).astype(np.float32)
        self.betas = torch.tensor(avg, requires_grad=True)
        self.betas_ref = torch.tensor(avg)
        self.valid = torch.tensor((valid_smpl > 0.7).astype(np.float32))
        self.zmin_lin = torch.tensor(np.ones_like(max_z), requires_grad=True)
        self.zmax_lin = torch.tensor(2.0 * max_z, requires_grad=True)
        self.scene_pcd = None
        self.scene_depth = None
        self.pT_filt = None
        self.v_filt = None
        return log

    def leaves(self):
        return [self.poses_T, self.poses_smpl, self.betas, self.zmin_lin, self.zmax_lin, self.xscale]

    # -- optimizer.py:605-616 ---------------------------------------------------------------
    def update_scene_pointcloud(self, depth, mask):
        gx = np.linspace(0.5, self.W - 0.5, self.W)
        gy = np.linspace(0.5, self.H - 0.5, self.H)
        grid = np.stack(np.meshgrid(gx, gy, indexing='xy'), axis=-1).astype(np.float32)
        uvd = torch.cat([torch.tensor(grid), torch.tensor(np.asarray(depth, np.float32))[..., None]], -1).view(1, -1, 3)
        xyz = unproject_points(uvd, self.Kt)[0]
        keep = torch.tensor(np.asarray(mask)).view(-1) > 0.5
        self.scene_pcd = xyz[keep][None, None]
        self.scene_depth = depth

    # -- optimizer.py:678-707 ---------------------------------------------------------------
    def _eval_batch(self, idx):
        b = len(idx)
        N = self.N
        scale = torch.pow(torch.tensor(1.1), self.xscale)
        min_z = softplus(self.zmin_lin[idx])
        max_z = min_z.detach() + 1.0 + softplus(self.zmax_lin[idx])
        th = self.poses_smpl[idx].view(-1, 72)
        be = self.betas.expand(b, N, 10).reshape(-1, 10)
        out = lbs_oracle.smpl_forward(self.model, be, th)
        verts = out['verts'].view(b, N, -1, 3)
        j17 = out[getattr(self, 'joints_key', 'joints_alphapose')].view(b, N, 17, 3)     # smpl_sparse_joints_key, optimizer.py:75, 695-696
        pT = self.poses_T[idx]
        verts_abs = scale * verts + pT
        ov = getattr(self, 'verts_value_override', None)
        if ov is not None:
            # test aid: evaluate everything downstream AT the vertices of the kernel under test (values replaced, the
            # gradient still flows through this oracle's own LBS).  The rasterised terms are ill-conditioned at sliver
            # faces seen edge-on (gradient ~ 1/area): a 1e-7 m difference between two fp32 LBS evaluations moves such a
            # body's gradient by percents, which says nothing about either implementation.
            verts_abs = verts_abs + (ov[idx].to(verts_abs.dtype) - verts_abs).detach()
        return dict(scale=scale, min_z=min_z, max_z=max_z, pT=pT,
                    verts=verts_abs, joints=scale * j17 + pT, poses=self.poses_smpl[idx])

    def full_sequence_verts(self):
        T, N = self.T, self.N
        be = self.betas.expand(T, N, 10).reshape(-1, 10)
        out = lbs_oracle.smpl_forward(self.model, be, self.poses_smpl.view(-1, 72))
        return torch.pow(torch.tensor(1.1), self.xscale) * out['verts'].view(T, N, -1, 3) + self.poses_T

    # -- one batch of optimizer.py:394-554 --------------------------------------------------
    def batch_loss(self, data):
        idx = data['idxs'].long()
        v = self._eval_batch(idx)
        b, N = len(idx), self.N
        W, H = self.W, self.H
        conf = (data['pose2d'][..., 2:3] >= self.thr).float()                     # :404
        p2d_valid = (conf.sum(dim=(2, 3)) >= 2).float()                           # :405
        mask_valid = (data['seg_mask'].sum(dim=(2, 3)) >= 0.005 * H * W).float()  # :407-409
        terms = {}
        # 2D joints :414-420
        uv = project_points(v['joints'].view(b * N, 17, 3), self.Kt.expand(b * N, 3, 3), self.Kd).view(b, N, 17, 2)
        nrm = torch.tensor([[[[float(W), float(H)]]]])
        w = self.joint_w * conf
        terms['loss_pose24j'] = ((w * uv / nrm - w * data['pose2d'][..., :2] / nrm) ** 2).sum()
        # depth + silhouette :425-477
        loss_depth = torch.zeros(())
        loss_sil = torch.zeros(())
        tgt_disp = data['depths'] * (1.0 / v['min_z'] - 1.0 / v['max_z']) + 1.0 / v['max_z']   # :425
        if self.rasteriser is not None:
            # (a rasteriser with ``wants_frames`` is also told which frames these bodies belong to: tests hand in the face
            # selection of the kernel under test, per frame)
            if getattr(self.rasteriser, 'wants_frames', False):
                zbuf, alpha = self.rasteriser(v['verts'].view(b * N, -1, 3), frames=idx)
            else:
                zbuf, alpha = self.rasteriser(v['verts'].view(b * N, -1, 3))
            zbuf = zbuf.view(b, N, H, W)
            alpha = alpha.view(b, N, H, W)
            er = erode3x3(erode3x3(data['seg_mask']))                             # :434
            m = (zbuf > 0).float() * er * p2d_valid[..., None, None]              # :432-438
            pred_disp = 1.0 / torch.clamp(zbuf + 0.2, min=self.eps)               # :440
            loss_depth = avg_log_depth_loss(pred_disp, tgt_disp.unsqueeze(1), m)          # :441-442
            order = torch.argsort(v['pT'][..., 0, 2], dim=1)                      # :450
            for j in range(b):
                acc = torch.zeros(H, W)
                for r in range(N):
                    n = int(order[j, r])
                    seg = data['seg_mask'][j, n]
                    # quirk kept: the gate is indexed by the RANK r (``nj``), not by the person
                    # picked for that rank (``nj_s``)                                :472-474
                    if float(mask_valid[j, r] * p2d_valid[j, r]) > 0:
                        loss_sil = loss_sil + masked_mse_loss(alpha[j, n], seg, 1 - acc)
                    acc = ((acc + seg) > 0).float()                               # :475
        terms['loss_depth'] = loss_depth
        terms['loss_silhouette'] = loss_sil
        # contact + foot sliding :485-518
        contact = torch.zeros(())
        foot = torch.zeros(())
        if self.scene_pcd is not None and self.scene_depth is not None:
            gv = v['verts']
            low_idx = torch.argmax(gv[..., 1], dim=2)                             # (b,N)  :487
            gi = low_idx[..., None, None].expand(b, N, 1, 3)
            low = torch.gather(gv, 2, gi)                                         # (b,N,1,3)
            d2 = ((self.scene_pcd - low) ** 2).sum(-1)                            # (b,N,M)  :492
            k = min(32, d2.shape[-1])
            nn = torch.argsort(d2, dim=-1)[..., :k]                               # :495
            pts = self.scene_pcd[0, 0][nn]                                        # (b,N,k,3)
            mean_pt = pts.mean(dim=2, keepdim=True)                               # :500
            dy = (mean_pt - low)[..., 1:2]                                        # :502
            tgt = v['pT'].detach().clone()
            tgt[..., 1:2] = tgt[..., 1:2] + (dy + 0.02).detach()                  # :504-505
            contact = (v['pT'] - tgt).abs().sum()                                 # :506
            gate = (dy > -0.20)[1:].float()                                       # :510-513
            low_t = low[1:]
            low_tm1 = torch.gather(gv[:-1], 2, gi[1:])                            # :514
            foot = (gate * low_t - gate * low_tm1).abs().sum() / torch.clamp(gate.sum(), min=1)   # :515-518
        terms['reg_contact'] = contact
        terms['reg_foot_sliding'] = foot
        # priors :523-532
        ref = (self.valid[idx] * data['poses_smpl'] - self.valid[idx] * v['poses']).abs().sum()
        ref = ref + b * (self.betas - self.betas_ref).abs().sum()
        terms['reg_ref_poses'] = ref
        s_avg = ((v['scale'] - 1.0).sum()) ** 2
        s_per = ((v['scale'] - 1.0) ** 2).mean()
        terms['reg_scale'] = s_avg + s_per
        c = self.c
        total = (c['proj2d'] * terms['loss_pose24j'] + c['depth'] * loss_depth + c['silhouette'] * loss_sil
                 + c['reg_poses'] * ref + c['reg_scales'] * s_per + float(c['reg_scales'] > 0) * s_avg
                 + c['reg_contact'] * contact + c['reg_foot_sliding'] * foot)     # :535-542
        return total, terms, tgt_disp

    def temporal_loss(self):
        """optimizer.py:560-575"""
        vel = ((self.poses_T[1:] - self.poses_T[:-1]) ** 2).sum()
        loss = self.c['reg_velocity'] * vel
        filt = torch.zeros(())
        if self.pT_filt is not None and self.v_filt is not None:
            gv = self.full_sequence_verts()
            filt = (((gv[1:] - gv[:-1]) - (self.v_filt[1:] - self.v_filt[:-1])) ** 2).sum()
            loss = loss + self.c['reg_verts_filter'] * filt
        return loss, vel, filt

    def update_filters(self, c1=0.01, b1=0.02, c2=0.001, b2=0.5, cpu_alias_quirk=False):
        """optimizer.py:383-392.

        ``cpu_alias_quirk``: on a CPU device the reference's ``x.cpu().detach().numpy()``
        (optimizer.py:665) shares storage with the leaf, so filtering ``poses_T`` OVERWRITES the
        leaf with its filtered values before the vertices are filtered (a side effect that does not
        exist on a "cuda" device, where ``.cpu()`` copies).  The golden fixtures come from the CPU
        run and are pinned with the flag on; the HIP path follows the device semantics (flag off).
        """
        with torch.no_grad():
            filt = torch.tensor(one_euro_sequence(self.poses_T.detach().numpy(), c1, b1))
            self.pT_filt = filt
            if cpu_alias_quirk:
                self.poses_T.copy_(filt)
            self.v_filt = torch.tensor(one_euro_sequence(self.full_sequence_verts().detach().numpy(), c2, b2))

    def cycle_grads(self, batches):
        """One cycle's accumulated gradients (no step): optimizer.py:376-575."""
        for p in self.leaves():
            p.grad = None
        logs = []
        for data in batches:
            total, terms, _ = self.batch_loss(data)
            total.backward()
            logs.append({k: float(x.detach()) for k, x in terms.items()})
        lt, vel, filt = self.temporal_loss()
        lt.backward()
        log = {k: float(np.mean([l[k] for l in logs])) for k in logs[0]}
        log['reg_vel'] = float(vel)
        log['reg_filter_verts'] = float(filt)
        return log

    def fit(self, batches, num_iter, update_filters_every=25, scene_update=None, cpu_alias_quirk=False):
        """optimizer.py:324-602 (scene update delegated to ``scene_update(self, cycle)``).  ``batches``: the list of
        batch dicts every cycle sees, or a callable cycle -> list (a shuffling dataloader delivers different batches
        every cycle, configs/predict_mupots.yml:14)."""
        leaves = self.leaves()
        sq = [torch.zeros_like(p) for p in leaves]
        buf = [torch.zeros_like(p) for p in leaves]
        lr = 0.01
        out = []
        for cycle in range(num_iter):
            if cycle >= 30 and cycle % update_filters_every == 0:
                self.update_filters(cpu_alias_quirk=cpu_alias_quirk)
            out.append(self.cycle_grads(batches(cycle) if callable(batches) else batches))
            if scene_update is not None and cycle >= 30:
                scene_update(self, cycle)
            with torch.no_grad():
                for p, s, bf in zip(leaves, sq, buf):
                    g = p.grad if p.grad is not None else torch.zeros_like(p)
                    rmsprop_step(p, g, s, bf, lr)
            lr *= 0.99
        return out

    def optimized_variables(self):
        """optimizer.py:619-636"""
        with torch.no_grad():
            min_z = softplus(self.zmin_lin)
            max_z = min_z + 1.0 + softplus(self.zmax_lin)
            return dict(scale_factor=torch.pow(torch.tensor(1.1), self.xscale).numpy(),
                        poses_T=self.poses_T.numpy().copy(), poses_smpl=self.poses_smpl.numpy().copy(),
                        betas_smpl=self.betas.numpy().copy(), valid_smpl=self.valid.numpy().copy(),
                        min_z=min_z.numpy(), max_z=max_z.numpy())
