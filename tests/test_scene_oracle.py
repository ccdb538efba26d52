"""Known-answer pins of oracle/scene_oracle.py, the checker of the device scene aggregation (csrc/mh_sceneagg.hip).

OpenCV is not in the image, so the three cv2 calls of the reference's ``postprocess_depthmap`` (utils.py:174-209:
``cv2.bilateralFilter(d=9, sigmaColor=.05, sigmaSpace=25)``, ``cv2.Sobel(ksize=3)``, ``cv2.erode(3x3, iterations=2)``) and
the python loops of ``fillin_values`` (utils.py:91-135) cannot be compared with their originals.  They are pinned here
on inputs whose exact output follows from the documented definitions -- stage by stage, then the whole pipeline.  The
same pipeline-level cases run against the HIP kernels in tests/test_scene_agg_gpu.py (``scene_known_answer_cases``)."""
import numpy as np
import pytest

from oracle import scene_oracle as so


# ---- stage level ---------------------------------------------------------------------------------------------------------
def test_bilateral_keeps_a_constant_image():
    x = np.full((20, 31), 0.37, np.float32)
    np.testing.assert_allclose(so._bilateral(x, 9, 0.05, 25), x, rtol=1e-7)


def test_bilateral_keeps_a_step_that_is_many_sigma_color_high():
    """range weight across the step = exp(-(0.3)^2 / (2 * 0.05^2)) = e^-18 = 1.5e-8: each side stays what it was"""
    x = np.full((24, 40), 0.5, np.float32)
    x[:, 20:] = 0.2
    np.testing.assert_allclose(so._bilateral(x, 9, 0.05, 25), x, rtol=2e-7)


def test_bilateral_is_the_gaussian_window_mean_when_sigma_color_is_huge():
    """with the range term switched off the filter is the normalised spatial Gaussian over the CIRCULAR support of radius 4
    (OpenCV skips taps with dy^2 + dx^2 > r^2), borders reflected without repeating the edge sample (BORDER_REFLECT_101)"""
    rng = np.random.RandomState(0)
    x = rng.uniform(0.1, 1.0, (15, 17)).astype(np.float32)
    got = so._bilateral(x, 9, 1e9, 25)
    p = np.pad(x.astype(np.float64), 4, mode='reflect')
    want = np.zeros_like(x, dtype=np.float64)
    for i in range(x.shape[0]):
        for j in range(x.shape[1]):
            num = den = 0.0
            for dy in range(-4, 5):
                for dx in range(-4, 5):
                    if dy * dy + dx * dx <= 16:
                        w = np.exp(-(dy * dy + dx * dx) / (2.0 * 25.0 ** 2))
                        num += w * p[i + 4 + dy, j + 4 + dx]
                        den += w
            want[i, j] = num / den
    np.testing.assert_allclose(got, want, rtol=1e-6)
    # a centred impulse far from the border spreads over exactly the 49 taps of the radius-4 disc
    imp = np.zeros((15, 17), np.float32)
    imp[7, 8] = 1.0
    assert int((so._bilateral(imp, 9, 1e9, 25) > 0).sum()) == 49


def test_sobel_of_a_ramp():
    """f = 3 + 0.25 x: d/dx through [-1 0 1] x [1 2 1] = 8 * 0.25 in the interior, 0 in the first / last column
    (REFLECT_101 mirrors x=1 into x=-1), d/dy = 0 everywhere"""
    H, W = 9, 12
    f = (3.0 + 0.25 * np.arange(W, dtype=np.float32))[None, :].repeat(H, 0)
    gx, gy = so._sobel(f, 1, 0), so._sobel(f, 0, 1)
    np.testing.assert_allclose(gx[:, 1:-1], 2.0, rtol=1e-6)
    np.testing.assert_array_equal(gx[:, 0], 0)
    np.testing.assert_array_equal(gx[:, -1], 0)
    np.testing.assert_array_equal(gy, 0)
    gyt = so._sobel(f.T.copy(), 0, 1)
    np.testing.assert_allclose(gyt[1:-1], 2.0, rtol=1e-6)


def test_erode_twice_grows_a_hole_by_two_and_leaves_the_border_alone():
    m = np.ones((12, 14), np.float32)
    m[6, 7] = 0
    m[0, 0] = 0
    e = so._erode3(m, 2)
    want = np.ones_like(m)
    want[4:9, 5:10] = 0          # 5x5 around (6,7)
    want[0:3, 0:3] = 0           # 3x3 clipped by the corner; the image border itself does not erode (+inf outside)
    np.testing.assert_array_equal(e, want)
    np.testing.assert_array_equal(so._erode3(np.ones((5, 5), np.float32), 2), np.ones((5, 5), np.float32))


def column_hole_case():
    """value = column index, columns 10..18 masked out: two sweeps of the 7x7 fill"""
    H, W = 16, 30
    x = np.arange(W, dtype=np.float32)[None, :].repeat(H, 0).copy()
    mask = np.ones((H, W), np.float32)
    mask[:, 10:19] = 0
    x[mask == 0] = -100.0                 # whatever sits under the mask must never be read
    # sweep 1 reads the valid values of its 7-wide window: {7,8,9} -> 8, {8,9} -> 8.5, {9} -> 9 | 19, {19,20} -> 19.5, 20;
    # sweep 2 reads sweep 1's results (never a value written in the same sweep): col 13 sees {8, 8.5, 9, 19} (equal
    # multiplicity per row) -> (8.5 + 9) / 2, col 14 {8.5, 9, 19, 19.5} -> 14, col 15 {9, 19, 19.5, 20} -> 19.25
    filled = np.array([8, 8.5, 9, 8.75, 14, 19.25, 19, 19.5, 20], np.float32)
    return x, mask, filled


def test_fill_sweeps():
    x, mask, filled = column_hole_case()
    x1, m1 = so.fillin_values(x, mask, filter_size=7)
    np.testing.assert_array_equal(m1[:, 10:19], np.array([1, 1, 1, 0, 0, 0, 1, 1, 1], np.float32)[None, :].repeat(16, 0))
    x2, m2 = so.fillin_values(x1, m1, filter_size=7)
    assert m2.min() == 1
    np.testing.assert_allclose(x2[:, 10:19], filled[None, :].repeat(16, 0), rtol=1e-6)
    np.testing.assert_array_equal(x2[:, :10], x[:, :10])
    np.testing.assert_array_equal(x2[:, 19:], x[:, 19:])


# ---- pipeline level (also run on the device) ---------------------------------------------------------------------------------
def scene_known_answer_cases():
    """(name, depth, mask, use_bilateral, expected, rtol) for postprocess_depthmap(depth, mask, 7, use_bilateral)"""
    cases = []
    H, W = 40, 64
    # 1. constant depth with a hole: zero gradients -> 0/0 statistics -> NaN > NaN is False -> no edges; the hole is
    #    filled with the only value there is
    d = np.full((H, W), 4.0, np.float32)
    m = np.ones((H, W), np.float32)
    m[10:22, 30:37] = 0
    din = d.copy()
    din[m == 0] = 0.0
    cases.append(('constant_with_hole', din, m, 1, d, 1e-6))
    # 2. a clean step 2 m | 5 m (disparity .5 | .2 = 6 sigma_color apart): the bilateral keeps both sides, Sobel fires
    #    on the two columns at the step, two erosions mask six columns, and the fill puts back the side each came from
    d = np.full((H, W), 2.0, np.float32)
    d[:, 32:] = 5.0
    cases.append(('step', d, np.ones((H, W), np.float32), 1, d.copy(), 1e-6))
    # 3. the same step on a gentle ramp, bilateral off: exactly columns c0-2 .. c0+3 (c0 = 31, the last near column) are
    #    re-filled -- two edge columns grown by two 3x3 erosions -- with the medians of the valid columns in reach
    x = np.arange(W, dtype=np.float32)
    row = np.where(x <= 31, 2.0 + 0.001 * x, 5.0 + 0.001 * x).astype(np.float32)
    d = row[None, :].repeat(H, 0).copy()
    want = d.copy()
    c0 = 31
    new = [row[c0 - 4], 0.5 * (row[c0 - 4] + row[c0 - 3]), row[c0 - 3], row[c0 + 4], 0.5 * (row[c0 + 4] + row[c0 + 5]), row[c0 + 5]]
    want[:, c0 - 2:c0 + 4] = np.array(new, np.float32)[None, :]
    cases.append(('step_on_ramp_no_bilateral', d, np.ones((H, W), np.float32), 0, want, 1e-6))
    return cases


@pytest.mark.parametrize('case', scene_known_answer_cases(), ids=lambda c: c[0])
def test_postprocess_known_answers(case):
    _, depth, mask, bil, want, rtol = case
    with np.errstate(invalid='ignore', divide='ignore'):
        got = so.postprocess_depthmap(depth, mask, fillin_ksize=7, use_bilateral_filter=bool(bil))
    np.testing.assert_allclose(got, want, rtol=rtol)
