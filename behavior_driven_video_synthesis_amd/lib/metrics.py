"""Evaluation hooks of the hot path's output on the device (reference ``lib/metrics.py:22-116``).

``compute_ssim`` there runs the model over a reconstruction loader, moves every batch to the host and calls
``skimage.metrics.structural_similarity(..., multichannel=True, data_range=1.0, gaussian_weights=True,
use_sample_covariance=False)`` image by image (:94-107).  Here the SSIM map, its Gaussian windows and the mean are
one HIP kernel over the whole batch (``vunet_ssim_partial``); the dataset / sampler plumbing is out of scope, so the
hook takes an iterable of batches.

The Inception-based scores (``compute_fid`` :119-282, ``_calculate_fid`` :285-322, ``inception_score`` :362-415) are
built as DRIVERS with a pluggable feature extractor / classifier: the pretrained torchvision Inception-v3 weights are
third-party data this build cannot fetch (like the VGG19 weights, they are user-supplied: any callable mapping an image
batch to features / logits).  The statistics -- feature mean and covariance, the matrix square root, traces, softmax and
the KL divergences -- are float64 on the host exactly as the reference computes them (numpy / scipy), pinned against
the reference's own functions on synthetic features (tests/golden/g7_metrics.npz).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Iterable, Optional, Tuple

import numpy as np
import torch

from . import utils
from .. import ops

_SIGMA, _TRUNCATE = 1.5, 3.5


def _window(device) -> torch.Tensor:
    r = int(_TRUNCATE * _SIGMA + 0.5)
    w = [math.exp(-0.5 * (i * i) / (_SIGMA * _SIGMA)) for i in range(-r, r + 1)]
    s = sum(w)
    return torch.tensor([v / s for v in w], dtype=torch.float32, device=device)


def ssim(rec: torch.Tensor, target: torch.Tensor, data_range: float = 1.0) -> torch.Tensor:
    """Per-image SSIM of two [N, C, H, W] batches with values in [0, data_range] -> [N] (fp32)."""
    if rec.shape != target.shape or rec.dim() != 4:
        raise ValueError("ssim expects two [N, C, H, W] tensors of the same shape")
    ops._dev(rec)
    ops._dev(target)
    rec, target = rec.contiguous().float(), target.contiguous().float()
    n, c, h, w = rec.shape
    tiles = ((h + 7) // 8) * ((w + 31) // 32)
    partial = torch.empty(n * c * tiles, device=rec.device, dtype=torch.float32)
    ops._call("vunet_ssim_partial", ops._p(rec), ops._p(target), n * c, h, w, float(data_range), ops._p(_window(rec.device)),
              ops._p(partial), ops._stream())
    return partial.view(n, c * tiles).sum(dim=1) / float(c * (h - 10) * (w - 10))


def psnr(rec: torch.Tensor, target: torch.Tensor, data_range: float = 1.0) -> torch.Tensor:
    """Per-image PSNR in dB -> [N]."""
    mse = (rec.float() - target.float()).pow(2).flatten(1).mean(dim=1)
    return 10.0 * torch.log10(data_range * data_range / mse)


@torch.no_grad()
def compute_ssim(model: torch.nn.Module, batches: Iterable[Dict[str, torch.Tensor]], max_n_samples: int = 8000,
                 inplane_normalize: bool = False) -> float:
    """lib/metrics.py:22-116 on an iterable of batches: reconstruct (``model(app, stickman)[0]``), scale both images
    to [0, 1] (:86-92) and average the per-image SSIM over at most ``max_n_samples`` samples."""
    was_training = model.training
    model.eval()
    total, count = None, 0
    for batch in batches:
        target = batch["pose_img"]
        if "app_img" in batch and type(model).__name__ == "VunetOrg":   # :71-74
            app = batch["app_img"]
        else:
            app = batch["pose_img_inplane"] if inplane_normalize else target
        rec = model(app, batch["stickman"])[0]
        vals = ssim(utils.scale_img(rec), utils.scale_img(target))
        take = min(vals.numel(), max_n_samples - count)
        s = vals[:take].sum()
        total = s if total is None else total + s
        count += take
        if count >= max_n_samples:
            break
    model.train(was_training)
    return float(total / count) if count else float("nan")


# ------------------------------------------------------------------------------------------------
# FID / Inception score (lib/metrics.py:119-415): drivers around a user-supplied Inception network
# ------------------------------------------------------------------------------------------------
def _calculate_fid(mu1, cov1, mu2, cov2, eps: float = 1e-6) -> float:
    """lib/metrics.py:285-322 (Frechet distance of two Gaussians):
    ``|mu1 - mu2|^2 + tr(S1) + tr(S2) - 2 tr(sqrtm(S1 S2))`` with the reference's handling of a singular product (eps on the
    diagonals) and of the small imaginary part ``scipy.linalg.sqrtm`` may return."""
    from scipy import linalg
    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(cov1), np.atleast_2d(cov2)
    assert mu1.shape == mu2.shape, "Training and test mean vectors have different lengths"
    assert sigma1.shape == sigma2.shape, "Training and test covariances have different dimensions"
    diff = mu1 - mu2
    covmean, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    if not np.isfinite(covmean).all():
        print(f"fid calculation produces singular product; adding {eps} to diagonal of cov estimates")
        offset = np.eye(sigma1.shape[0]) * eps
        covmean = linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
    if np.iscomplexobj(covmean):
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
            raise ValueError(f"Imaginary component {np.max(np.abs(covmean.imag))}")
        covmean = covmean.real
    return float(diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean))


def fid_from_features(gt_features: np.ndarray, gen_features: np.ndarray) -> float:
    """:211-212, :272-275: mean and ``np.cov(rowvar=False)`` of each [N, D] feature set, then ``_calculate_fid``."""
    gt, gen = np.asarray(gt_features, dtype=np.float64), np.asarray(gen_features, dtype=np.float64)
    return _calculate_fid(np.mean(gt, axis=0), np.cov(gt, rowvar=False), np.mean(gen, axis=0), np.cov(gen, rowvar=False))


@torch.no_grad()
def compute_fid(model: torch.nn.Module, batches: Iterable[Dict[str, torch.Tensor]], feature_extractor: Callable,
                gt_features: Optional[np.ndarray] = None, max_n_samples: int = 8000,
                inplane_normalize: bool = False) -> float:
    """lib/metrics.py:119-282 on an iterable of batches.  ``feature_extractor(images) -> [N, D]`` stands for the
    reference's ``FIDInceptionModel`` (user-supplied pretrained Inception-v3 pool features); ground-truth features are
    computed from ``batch["pose_img"]`` unless ``gt_features`` (the reference's cached ``<dataset>-fid-features.npy``) is
    given; generated features come from the reconstructions ``model(app, stickman)[0]`` (:243-258)."""
    was_training = model.training
    model.eval()
    gts, gens, n = [], [], 0
    for batch in batches:
        target = batch["pose_img"]
        if "app_img" in batch and type(model).__name__ == "VunetOrg":
            app = batch["app_img"]
        else:
            app = batch["pose_img_inplane"] if inplane_normalize else target
        if gt_features is None:
            gts.append(feature_extractor(target).detach().double().cpu().numpy())
        gens.append(feature_extractor(model(app, batch["stickman"])[0]).detach().double().cpu().numpy())
        n += target.shape[0]
        if n >= max_n_samples:
            break
    model.train(was_training)
    gt = np.asarray(gt_features) if gt_features is not None else np.concatenate(gts, axis=0)
    return fid_from_features(gt, np.concatenate(gens, axis=0))


def inception_score_from_probs(preds: np.ndarray, splits: int = 1) -> Tuple[float, float]:
    """:403-415: per split, ``exp(mean_i KL(p(y|x_i) || p(y)))`` with ``p(y)`` the split's mean prediction
    (``scipy.stats.entropy(pyx, py)``); returns (mean, std) over the splits."""
    from scipy.stats import entropy
    preds = np.asarray(preds, dtype=np.float64)
    n = preds.shape[0]
    split_scores = []
    for k in range(splits):
        part = preds[k * (n // splits): (k + 1) * (n // splits), :]
        py = np.mean(part, axis=0)
        split_scores.append(np.exp(np.mean([entropy(part[i, :], py) for i in range(part.shape[0])])))
    return float(np.mean(split_scores)), float(np.std(split_scores))


@torch.no_grad()
def inception_score(imgs, classifier: Callable, batch_size: int = 32, resize: bool = False, splits: int = 1,
                    device=None) -> Tuple[float, float]:
    """lib/metrics.py:362-415.  ``imgs``: tensor [N, 3, H, W] in [-1, 1] (or a dataset of 1-tuples of such images, as the
    reference takes); ``classifier(images) -> logits [n, K]`` stands for torchvision's pretrained ``inception_v3``
    (user-supplied).  ``resize``: bilinear up-sampling to 299 x 299 first (:385, ``nn.Upsample(size=(299, 299),
    mode='bilinear')`` -- align_corners False)."""
    if not torch.is_tensor(imgs):
        imgs = torch.stack([imgs[i][0] if isinstance(imgs[i], (tuple, list)) else imgs[i] for i in range(len(imgs))])
    n = imgs.shape[0]
    assert batch_size > 0 and n > batch_size
    preds = []
    for i in range(0, n, batch_size):
        x = imgs[i:i + batch_size]
        if device is not None:
            x = x.to(device)
        if resize:
            x = torch.nn.functional.interpolate(x.float(), size=(299, 299), mode="bilinear", align_corners=False)
        preds.append(torch.softmax(classifier(x).double(), dim=1).cpu().numpy())
    return inception_score_from_probs(np.concatenate(preds, axis=0), splits)
