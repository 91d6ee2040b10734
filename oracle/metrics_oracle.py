"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the evaluation statistics of the reference's lib/metrics.py: the
SSIM its hook computes (below), the Frechet distance of ``_calculate_fid`` (:285-322) and the Inception score's split
statistic (:403-415).  The FID / IS restatements are PINNED: tests/golden/g7_metrics.npz holds the outputs of the
reference's own functions on seeded synthetic features (tests/test_metrics_oracle.py).  They deliberately take a
different numerical route than the product (eigenvalues of S1 S2 instead of scipy's sqrtm; explicit sums instead of
scipy.stats.entropy), so agreement is a real check.

SSIM:

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


def scale_img(x):
    """lib/utils.py:658-668: [-1, 1] -> [0, 1], clamped (:666-667)."""
    return np.clip((np.asarray(x, dtype=np.float64) + 1.0) / 2.0, 0.0, 1.0)


def frechet_distance(mu1, cov1, mu2, cov2) -> float:
    """|mu1 - mu2|^2 + tr(S1) + tr(S2) - 2 tr((S1 S2)^(1/2)); tr of the square root = sum of the square roots of the
    eigenvalues of S1 S2 (real and >= 0 for two covariance matrices, up to rounding)."""
    mu1, mu2 = np.asarray(mu1, dtype=np.float64), np.asarray(mu2, dtype=np.float64)
    s1, s2 = np.asarray(cov1, dtype=np.float64), np.asarray(cov2, dtype=np.float64)
    ev = np.linalg.eigvals(s1 @ s2)
    tr_sqrt = float(np.sum(np.sqrt(np.clip(ev.real, 0.0, None))))
    d = mu1 - mu2
    return float(d @ d + np.trace(s1) + np.trace(s2) - 2.0 * tr_sqrt)


def fid_from_features(a, b) -> float:
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)

    def stats(f):
        mu = f.mean(axis=0)
        c = (f - mu).T @ (f - mu) / (f.shape[0] - 1)
        return mu, c
    (m1, c1), (m2, c2) = stats(a), stats(b)
    return frechet_distance(m1, c1, m2, c2)


def inception_score_from_probs(p, splits=1):
    p = np.asarray(p, dtype=np.float64)
    n = p.shape[0]
    scores = []
    for k in range(splits):
        part = p[k * (n // splits):(k + 1) * (n // splits)]
        q = part / part.sum(axis=1, keepdims=True)
        py = part.mean(axis=0)
        py = py / py.sum()
        kl = np.sum(np.where(q > 0, q * (np.log(np.where(q > 0, q, 1.0)) - np.log(py)), 0.0), axis=1)
        scores.append(np.exp(kl.mean()))
    return float(np.mean(scores)), float(np.std(scores))
