"""Pin the INSIDE of the rasteriser to PyTorch3D -- to be run wherever PyTorch3D is installed.

The reference's two rasterised terms call PyTorch3D (optimizer.py:211-232 builds the cameras and the two
RasterizationSettings, :428-431 and :447-448 call them).  PyTorch3D is in neither the build container nor the GPU image, so
everything in this repository that stands for it -- oracle/raster_oracle.py, oracle/raster_select.c and the HIP kernels
checked against them -- is a restatement of its published naive rasteriser that no PyTorch3D output pins: "parity unpinned"
(DESIGN.md section 6).  This script produces the fixture that removes the caveat; tests/test_oracle_golden_pytorch3d.py (CPU:
the oracle) and tests/test_pytorch3d_golden_gpu.py (the HIP path) pick it up as soon as it exists.

    pip install pytorch3d      # any build; CPU is enough (the naive rasteriser is what the restatement follows)
    python tests/golden/make_golden_pytorch3d.py    ->   tests/golden/reference_pytorch3d.npz

It needs NEITHER /root/reference NOR a GPU: the scenes are the ones the committed raster fixtures already hold
(reference_raster_cpu.npz: two bodies that cross in depth on a 96x60 image, 20 frames; reference_raster_aspect_cpu.npz:
portrait and square images), posed by oracle/lbs_oracle.py (pinned to the reference's own SMPL by reference_cpu.npz), and
the cameras / settings are built exactly like optimizer.py:204-232 builds them.  Recorded per scene, for every body:
  zbuf8      (B,H,W)    fragments.zbuf[..., 0] of the K = 8, blur 1e-4 pass   -- all the depth term reads (:430)
  p2f8_0     (B,H,W)    its pix_to_face[..., 0] (face index inside the body's mesh, -1 = none)
  p2f4, z4, d4 (B,H,W,4) pix_to_face / zbuf / dists of the K = 4, blur 2e-5 pass (the silhouette renderer's fragments)
  alpha      (B,H,W)    SoftSilhouetteShader output [..., 3]                   -- all the silhouette term reads (:448)
  g_verts    (B,V,3)    d/dverts of  sum(wz * zbuf8) + sum(wa * alpha)  with the fixed weights wz, wa stored beside it
Numbers only; nothing of PyTorch3D's or the reference's source enters the file."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (HERE, ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def calibration(znear, zfar, K, size):
    """the 4x4 projection the reference hands to FoVPerspectiveCameras (transforms.py:222-255), re-derived"""
    W, H = size
    fx, fy, cx, cy = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
    if W > H:
        s, w1, h1 = 2 * fy / H, (W / H) * (W - 2 * cx) / W, (H - 2 * cy) / H
    elif H > W:
        s, w1, h1 = 2 * fx / W, (W - 2 * cx) / W, (H / W) * (H - 2 * cy) / H
    else:
        s, w1, h1 = 2 * (fx + fy) / (W + H), (W - 2 * cx) / W, (H - 2 * cy) / H
    f1, f2 = zfar / (zfar - znear), -(zfar * znear) / (zfar - znear)
    return np.array([[s, 0, w1, 0], [0, s, h1, 0], [0, 0, f1, f2], [0, 0, 1, 0]], np.float32)


def scenes():
    """[(name, verts (B,V,3) float32, faces (F,3) int64, cam_K (3,3), (W, H))] from the committed fixtures"""
    from mhhip import synthetic
    from oracle import lbs_oracle as lo
    import aspect_inputs as ai
    struct = synthetic.make_smpl_struct(1)
    regs = synthetic.make_extra_regressors(1, struct)
    model = lo.BodyModel(struct, regs)
    faces = np.asarray(struct.f).astype(np.int64)

    def posed(gr, frames):
        T, N, H, W = [int(x) for x in gr['in_dims']]
        with torch.no_grad():
            be = torch.tensor(gr['init_betas_smpl']).expand(T, N, 10).reshape(-1, 10)
            v = lo.smpl_forward(model, be, torch.tensor(gr['init_poses_smpl']).view(-1, 72))['verts'].view(T, N, -1, 3)
            v = torch.pow(torch.tensor(1.1), torch.tensor(gr['init_xscale_factor'])) * v + torch.tensor(gr['init_poses_T'])
        frames = [f for f in frames if f < T]
        return v[frames].reshape(-1, v.shape[2], 3).numpy().astype(np.float32), np.asarray(gr['in_cam_K'], np.float32), (W, H)
    out = []
    gr = dict(np.load(os.path.join(HERE, 'reference_raster_cpu.npz')))
    v, K, size = posed(gr, [0, 7, 9, 10, 11, 19])          # before, around and after the two bodies cross in depth
    out.append(('cross', v, faces, K, size))
    for tag in ai.VARIANTS:                                 # portrait and square images (the two other calibration branches)
        v, K, size = posed(ai.load(tag), [0, 2, 5])
        out.append((tag, v, faces, K, size))
    return out


def main():
    from pytorch3d.renderer import (FoVPerspectiveCameras, MeshRasterizer, MeshRenderer, RasterizationSettings,
                                    SoftSilhouetteShader)
    from pytorch3d.structures import Meshes
    import pytorch3d
    dev = torch.device('cpu')
    R = torch.tensor([[[-1., 0., 0.], [0., -1., 0.], [0., 0., 1.]]])
    Tt = torch.zeros(1, 3)
    out = {'pytorch3d_version': np.array(pytorch3d.__version__)}
    rng = np.random.RandomState(7)
    for name, verts, faces, K, (W, H) in scenes():
        cam = FoVPerspectiveCameras(R=R, T=Tt, K=torch.tensor(calibration(1.0, 100.0, K, (W, H))[None]), device=dev)
        s8 = RasterizationSettings(image_size=(H, W), blur_radius=1e-4, faces_per_pixel=8, perspective_correct=False)
        s4 = RasterizationSettings(image_size=(H, W), blur_radius=2e-5, faces_per_pixel=4, perspective_correct=False)
        rast8 = MeshRasterizer(cameras=cam, raster_settings=s8)
        rend4 = MeshRenderer(rasterizer=MeshRasterizer(cameras=cam, raster_settings=s4), shader=SoftSilhouetteShader())
        B, V = verts.shape[0], verts.shape[1]
        tv = torch.tensor(verts, requires_grad=True)
        tf = torch.tensor(faces)[None].expand(B, -1, -1)
        meshes = Meshes(tv, tf)
        fr8 = rast8(meshes)
        fr4 = rend4.rasterizer(meshes)
        alpha = rend4(meshes)[..., 3]
        wz = rng.uniform(0.5, 1.5, (B, H, W)).astype(np.float32)
        wa = rng.uniform(0.5, 1.5, (B, H, W)).astype(np.float32)
        zb = fr8.zbuf[..., 0]
        (torch.tensor(wz) * torch.where(zb > 0, zb, torch.zeros_like(zb))).sum().add((torch.tensor(wa) * alpha).sum()).backward()
        F = faces.shape[0]

        def local(p2f):                        # packed face index -> index inside the body's own mesh
            p = p2f.numpy().astype(np.int64)
            return np.where(p >= 0, p - (np.arange(B, dtype=np.int64) * F).reshape(B, *([1] * (p.ndim - 1))), -1)
        out.update({name + '_verts': verts, name + '_cam_K': np.asarray(K, np.float32), name + '_size': np.array([W, H], np.int64),
                    name + '_zbuf8': zb.detach().numpy(), name + '_p2f8_0': local(fr8.pix_to_face[..., 0]),
                    name + '_p2f4': local(fr4.pix_to_face), name + '_z4': fr4.zbuf.detach().numpy(),
                    name + '_d4': fr4.dists.detach().numpy(), name + '_alpha': alpha.detach().numpy(),
                    name + '_wz': wz, name + '_wa': wa, name + '_g_verts': tv.grad.numpy()})
        print(name, 'bodies', B, 'covered pixels', int((zb > 0).sum()))
    out['scene_names'] = np.array([s[0] for s in scenes()])
    np.savez_compressed(os.path.join(HERE, 'reference_pytorch3d.npz'), **out)
    print('wrote', os.path.join(HERE, 'reference_pytorch3d.npz'))


if __name__ == '__main__':
    main()
