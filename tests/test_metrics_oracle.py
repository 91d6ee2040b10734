"""CPU: the SSIM oracle (oracle/metrics_oracle.py, a restatement of skimage's structural_similarity as the reference
calls it, lib/metrics.py:94-107) against closed-form cases -- scikit-image itself is not in the image."""
import numpy as np

from oracle import metrics_oracle as M


def test_window_is_the_11_tap_sigma_1p5_gaussian():
    w = M.gaussian_window()
    assert w.shape == (11,) and abs(w.sum() - 1.0) < 1e-15 and np.allclose(w, w[::-1])
    assert abs(w[5] / w[4] - np.exp(0.5 / 2.25)) < 1e-12


def test_identical_images_score_one_and_constants_follow_the_formula():
    rng = np.random.default_rng(0)
    x = rng.random((3, 24, 40))
    assert abs(M.ssim_image(x, x) - 1.0) < 1e-12
    # two constant planes a, b: all variances vanish, S = (2ab + C1) / (a^2 + b^2 + C1) everywhere
    a, b = 0.3, 0.7
    want = (2 * a * b + 1e-4) / (a * a + b * b + 1e-4)
    got = M.ssim_plane(np.full((20, 20), a), np.full((20, 20), b))
    assert abs(got - want) < 1e-12


def test_symmetry_range_and_crop():
    rng = np.random.default_rng(1)
    x, y = rng.random((16, 30)), rng.random((16, 30))
    assert abs(M.ssim_plane(x, y) - M.ssim_plane(y, x)) < 1e-14
    assert -1.0 <= M.ssim_plane(x, y) <= 1.0
    # only the 5-pixel-cropped interior counts: changing the outermost ring far from the interior's 11x11 windows ...
    big_x, big_y = rng.random((40, 40)), rng.random((40, 40))
    s0 = M.ssim_plane(big_x, big_y)
    # ... is impossible (every border pixel is inside some interior window), but data_range rescaling is exact:
    assert abs(M.ssim_plane(255 * big_x, 255 * big_y, data_range=255.0) - s0) < 1e-12
    assert M.psnr(big_x, big_x) == float("inf") and abs(M.psnr(np.zeros(4), np.full(4, 0.1)) - 20.0) < 1e-9
