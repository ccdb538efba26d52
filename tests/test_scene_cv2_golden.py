"""The scene update's OpenCV primitives against OpenCV itself -- ACTIVE ONLY when tests/golden/reference_cv2.npz exists.

``cv2`` is installed neither in the build container nor on the GPU image: oracle/scene_oracle.py restates
``cv2.bilateralFilter`` / ``cv2.Sobel`` / ``cv2.erode`` (reference utils.py:174-209) from their documented semantics and is
pinned analytically (tests/test_scene_oracle.py).  tests/golden/make_golden_cv2.py records OpenCV's own outputs on any machine
that has it; with the file in place these tests hold the restatement (and, when the reference was importable there, the whole
``postprocess_depthmap``) against it.  Without the file: the generator's inputs are checked to be what the tests expect."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import scene_oracle as so

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = os.path.join(HERE, 'golden', 'reference_cv2.npz')
needs_fixture = pytest.mark.skipif(not os.path.exists(FIX), reason='tests/golden/reference_cv2.npz not generated yet '
                                   '(python tests/golden/make_golden_cv2.py where OpenCV is installed)')


def _generator():
    spec = importlib.util.spec_from_file_location('make_golden_cv2', os.path.join(HERE, 'golden', 'make_golden_cv2.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_generator_inputs_have_steps_holes_and_a_masked_block():
    for depth, mask in _generator().inputs():
        assert depth.dtype == np.float32 and mask.dtype == np.float32 and depth.shape == mask.shape
        assert np.isfinite(depth).all() and depth.min() >= 0.3 and depth.max() <= 20.0
        assert 0.005 < (mask == 0).mean() < 0.2
        assert np.abs(np.diff(depth, axis=1)).max() > 0.5                 # a depth step for the edge detector
        out = so.postprocess_depthmap(depth, mask, fillin_ksize=7, use_bilateral_filter=True)     # the restatement runs on them
        assert out.shape == depth.shape and np.isfinite(out).all()


@pytest.fixture(scope='module')
def cvf():
    return dict(np.load(FIX))


@needs_fixture
@pytest.mark.parametrize('i', [0, 1, 2, 3])
def test_restated_primitives_match_opencv(cvf, i):
    g = lambda k: cvf['m%d_%s' % (i, k)]
    depth = g('depth')
    bil = so._bilateral((1.0 / np.clip(depth, 0.01, 100)).astype(np.float32), 9, 0.05, 25)
    np.testing.assert_allclose(bil, g('bilateral'), rtol=2e-5, atol=1e-7)
    d2 = (1.0 / np.clip(g('bilateral'), 0.01, 100)).astype(np.float32)
    disp = (1.0 / np.clip(d2, 0.1, 100)).astype(np.float32)
    for nm, src in (('disp', disp), ('depth', d2)):
        np.testing.assert_allclose(so._sobel(src, 1, 0), g('sobel_%s_x' % nm), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(so._sobel(src, 0, 1), g('sobel_%s_y' % nm), rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(so._erode3(1 - g('edges'), 2), g('eroded'))


@needs_fixture
@pytest.mark.parametrize('i', [0, 1, 2, 3])
def test_whole_postprocess_matches_the_reference_with_opencv(cvf, i):
    key = 'm%d_ref_post' % i
    if key not in cvf:
        pytest.skip('the fixture was generated without /root/reference: primitives only')
    got = so.postprocess_depthmap(cvf['m%d_depth' % i], cvf['m%d_mask' % i], fillin_ksize=7, use_bilateral_filter=True)
    want = cvf[key]
    # the edge threshold (3 x mean of the gradient measure) may flip a borderline pixel and with it its fill: all but a
    # handful of pixels to 1e-5
    assert (np.abs(got - want) <= 1e-5 * np.maximum(1.0, np.abs(want))).mean() > 0.998
