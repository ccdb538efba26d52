"""Contact term neighbours (optimizer.py:591-603 -> pytorch3d.ops.knn_points K=32): the brute-force
scan and the uniform-grid search must both return the mean height of exactly the 32 nearest cloud
points, for surface-like and volumetric clouds, queries outside the cloud and clouds smaller than K."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dy_reference(pts, q, k):
    d = ((q[:, None, :].astype(np.float64) - pts[None].astype(np.float64)) ** 2).sum(-1)
    kk = min(k, pts.shape[0])
    idx = np.argsort(d, axis=1, kind='stable')[:, :kk]
    return pts[idx, 1].astype(np.float64).mean(1) - q[:, 1]


def _run(pts, q, k=32):
    from mhhip import _lib
    from mhhip._lib import check, ptr
    L = _lib.lib()
    dev = torch.device('cuda:0')
    tp = torch.tensor(pts, device=dev)
    tq = torch.tensor(q, device=dev)
    st = _lib.stream_ptr(dev)
    M, B = pts.shape[0], q.shape[0]
    brute = torch.empty(B, device=dev)
    check(L.mh_contact_knn(ptr(tp), M, ptr(tq), B, k, ptr(brute), st))
    ws = torch.empty(L.mh_scene_grid_bytes(M), dtype=torch.uint8, device=dev)
    check(L.mh_scene_grid_build(ptr(tp), M, ptr(ws), st))
    grid = torch.empty(B, device=dev)
    check(L.mh_contact_knn_grid(ptr(ws), M, ptr(tq), B, k, ptr(grid), st))
    torch.cuda.synchronize()
    return brute.cpu().numpy(), grid.cpu().numpy()


@pytest.mark.parametrize('kind', ['ground', 'volume', 'two_sheets', 'tiny', 'fewer_than_k'])
def test_grid_and_brute_force_neighbours(kind):
    rng = np.random.RandomState(7)
    if kind == 'ground':
        M = 23000
        pts = np.stack([rng.uniform(-5, 5, M), 1.2 + 0.02 * rng.randn(M) + 0.05 * np.sin(rng.uniform(0, 6, M)),
                        rng.uniform(2, 12, M)], 1)
    elif kind == 'volume':
        M = 40000
        pts = rng.uniform(-2, 2, (M, 3)) + np.array([0, 0, 5.0])
    elif kind == 'two_sheets':
        M = 30000
        pts = np.stack([rng.uniform(-4, 4, M), np.where(rng.rand(M) < 0.5, 1.0, -0.5) + 0.01 * rng.randn(M),
                        rng.uniform(3, 9, M)], 1)
    elif kind == 'tiny':
        M = 100
        pts = rng.uniform(-1, 1, (M, 3))
    else:
        M = 9
        pts = rng.uniform(-1, 1, (M, 3))
    pts = pts.astype(np.float32)
    B = 257
    q = pts[rng.randint(0, M, B)] + rng.randn(B, 3).astype(np.float32) * 0.3
    q[:16] += np.array([30.0, -20.0, 50.0], np.float32)          # far outside the cloud
    q = q.astype(np.float32)
    ref = _dy_reference(pts, q, 32)
    brute, grid = _run(pts, q)
    np.testing.assert_allclose(brute, ref, atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(grid, ref, atol=2e-5, rtol=1e-5)


def test_grid_is_rebuilt_in_place_for_a_new_cloud():
    rng = np.random.RandomState(3)
    q = rng.uniform(-1, 1, (64, 3)).astype(np.float32)
    for M in (5000, 700):
        pts = rng.uniform(-1, 1, (M, 3)).astype(np.float32) * np.array([3, 0.05, 3], np.float32)
        brute, grid = _run(pts, q)
        np.testing.assert_allclose(grid, _dy_reference(pts, q, 32), atol=2e-5, rtol=1e-5)
