"""Evaluation hooks of the hot path's output on the device (reference ``lib/metrics.py:22-116``).

``compute_ssim`` there runs the model over a reconstruction loader, moves every batch to the host and calls
``skimage.metrics.structural_similarity(..., multichannel=True, data_range=1.0, gaussian_weights=True,
use_sample_covariance=False)`` image by image (:94-107).  Here the SSIM map, its Gaussian windows and the mean are
one HIP kernel over the whole batch (``vunet_ssim_partial``); the dataset / sampler plumbing is out of scope, so the
hook takes an iterable of batches.  The Inception-based scores (FID / IS, :119-415) need pretrained Inception
weights and are not part of this build.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable

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
