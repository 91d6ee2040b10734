"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the SSIM the reference's evaluation hook computes.

lib/metrics.py:94-107 calls ``skimage.metrics.structural_similarity(rimg, timg, multichannel=True, data_range=1.0,
gaussian_weights=True, use_sample_covariance=False)`` on HWC float images in [0, 1].  scikit-image is not in this
image, so this file restates its published algorithm (Wang et al. 2004 as implemented in scikit-image 0.16-0.19,
``skimage/metrics/_structural_similarity.py``) on top of ``scipy.ndimage.gaussian_filter`` -- the very filter
skimage calls -- in float64, as skimage does.  **Parity unpinned** against skimage itself (absent); pinned by the
closed-form cases in tests/test_metrics_oracle.py.
"""
import numpy as np
from scipy.ndimage import gaussian_filter

K1, K2, SIGMA, TRUNCATE = 0.01, 0.03, 1.5, 3.5


def gaussian_window():
    """The 11 normalised taps scipy's gaussian_filter uses for sigma 1.5, truncate 3.5 (radius int(3.5*1.5+0.5) = 5)."""
    radius = int(TRUNCATE * SIGMA + 0.5)
    x = np.arange(-radius, radius + 1, dtype=np.float64)
    w = np.exp(-0.5 * x * x / (SIGMA * SIGMA))
    return w / w.sum()


def ssim_plane(x: np.ndarray, y: np.ndarray, data_range: float = 1.0) -> float:
    x, y = x.astype(np.float64), y.astype(np.float64)

    def f(a):
        return gaussian_filter(a, SIGMA, truncate=TRUNCATE)   # mode="reflect", as skimage leaves it
    ux, uy = f(x), f(y)
    uxx, uyy, uxy = f(x * x), f(y * y), f(x * y)
    vx, vy, vxy = uxx - ux * ux, uyy - uy * uy, uxy - ux * uy          # cov_norm = 1 (use_sample_covariance=False)
    c1, c2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
    pad = (2 * int(TRUNCATE * SIGMA + 0.5) + 1 - 1) // 2
    return float(s[pad:-pad, pad:-pad].mean())                          # crop(S, pad).mean()


def ssim_image(rec_chw: np.ndarray, tgt_chw: np.ndarray, data_range: float = 1.0) -> float:
    """multichannel=True: the mean of the per-channel values."""
    return float(np.mean([ssim_plane(r, t, data_range) for r, t in zip(rec_chw, tgt_chw)]))


def psnr(a: np.ndarray, b: np.ndarray, data_range: float = 1.0) -> float:
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return float("inf") if mse == 0 else 10.0 * np.log10(data_range ** 2 / mse)
