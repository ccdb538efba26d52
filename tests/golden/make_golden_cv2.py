"""Pin the OpenCV primitives inside the scene update -- to be run wherever ``cv2`` is installed.

``postprocess_depthmap`` (reference utils.py:174-209, called from optimizer.py:582) is three OpenCV calls around numpy:
``cv2.bilateralFilter(disp, 9, sigmaColor=0.05, sigmaSpace=25)``, ``cv2.Sobel(., CV_32F, 1|0, 0|1, ksize=3)`` on disparity and
depth, ``cv2.erode(1 - edges, ones((3,3)), iterations=2)``.  OpenCV is in neither the build container nor the GPU image, so
oracle/scene_oracle.py (and the device kernels of csrc/mh_sceneagg.hip checked against it) restate those three from their
documented semantics -- BORDER_REFLECT_101, circular bilateral support, +inf erosion border -- pinned analytically only
(tests/test_scene_oracle.py).  This script records what OpenCV itself returns; tests/test_scene_cv2_golden.py picks the file
up as soon as it exists (the oracle on the CPU, the device kernels under ``-m gpu``).

    pip install opencv-python-headless
    python tests/golden/make_golden_cv2.py      ->   tests/golden/reference_cv2.npz

Needs neither /root/reference nor a GPU.  With /root/reference present (and cv2 importable) the reference's own
``postprocess_depthmap`` is recorded as well (``ref_post_*``): then the composition is pinned by the reference itself, not
only its primitives.  Inputs: four depth maps of a ground plane + wall + boxes with holes / noise / a masked region (seeded
numpy; stored in the file).  Numbers only."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def inputs():
    out = []
    for seed, (H, W) in enumerate([(60, 96), (135, 240), (90, 54), (64, 64)]):
        rng = np.random.RandomState(100 + seed)
        ys = (np.arange(H, dtype=np.float32) + 0.5 - 0.45 * H) / (0.9 * H)
        d = np.minimum(np.where(ys[:, None] > 1e-3, 1.15 / np.maximum(ys[:, None], 1e-3), 10.0), 10.0)
        d = np.tile(d, (1, W)).astype(np.float32)
        for _ in range(4):                                  # boxes nearer than the background: depth steps
            y0, x0 = rng.randint(0, H - 8), rng.randint(0, W - 8)
            h, w = rng.randint(4, H // 2), rng.randint(4, W // 2)
            d[y0:y0 + h, x0:x0 + w] = np.minimum(d[y0:y0 + h, x0:x0 + w], rng.uniform(1.5, 6.0))
        d += rng.normal(0, 0.01, d.shape).astype(np.float32)
        d = np.clip(d, 0.3, 20.0).astype(np.float32)
        mask = np.ones((H, W), np.float32)
        mask[rng.randint(0, H - 6):, :][:6, rng.randint(0, W - 10):][:, :10] = 0      # a masked block
        mask[rng.rand(H, W) < 0.01] = 0                                               # and sprinkled holes
        out.append((d, mask))
    return out


def main():
    import cv2
    out = {'cv2_version': np.array(cv2.__version__)}
    ref_post = None
    if os.path.isdir('/root/reference'):
        sys.path.insert(0, '/root/reference')
        try:
            from mhmocap.utils import postprocess_depthmap as ref_post      # noqa: F401  (needs only numpy + cv2)
        except Exception as ex:       # other imports of that module missing: primitives only
            print('reference postprocess_depthmap not importable (%s): primitives only' % ex)
            ref_post = None
    for i, (depth, mask) in enumerate(inputs()):
        disp_in = (1.0 / np.clip(depth, 0.01, 100)).astype(np.float32)
        bil = cv2.bilateralFilter(disp_in, 9, sigmaColor=0.05, sigmaSpace=25)
        d2 = (1.0 / np.clip(bil, 0.01, 100)).astype(np.float32)
        disp = (1.0 / np.clip(d2, 0.1, 100)).astype(np.float32)
        sob = {}
        for nm, src in (('disp', disp), ('depth', d2)):
            sob[nm + '_x'] = cv2.Sobel(src, cv2.CV_32F, 1, 0, ksize=3)
            sob[nm + '_y'] = cv2.Sobel(src, cv2.CV_32F, 0, 1, ksize=3)
        sd = np.abs(sob['disp_x']) + np.abs(sob['disp_y'])
        sz = np.abs(sob['depth_x']) + np.abs(sob['depth_y'])
        grad = sd / sd.std() + sz / sz.std()
        edges = (grad > 3 * grad.mean()).astype(np.float32)
        er = cv2.erode((1 - edges), np.ones((3, 3)), iterations=2)
        pre = 'm%d_' % i
        out.update({pre + 'depth': depth, pre + 'mask': mask, pre + 'bilateral': bil, pre + 'sobel_disp_x': sob['disp_x'],
                    pre + 'sobel_disp_y': sob['disp_y'], pre + 'sobel_depth_x': sob['depth_x'], pre + 'sobel_depth_y': sob['depth_y'],
                    pre + 'edges': edges, pre + 'eroded': er})
        if ref_post is not None:
            out[pre + 'ref_post'] = ref_post(depth.copy(), mask.copy(), fillin_ksize=7, use_bilateral_filter=True).astype(np.float32)
        print('map', i, depth.shape, 'edge pixels', int(edges.sum()))
    out['count'] = np.array(len(inputs()))
    np.savez_compressed(os.path.join(HERE, 'reference_cv2.npz'), **out)
    print('wrote', os.path.join(HERE, 'reference_cv2.npz'))


if __name__ == '__main__':
    main()
