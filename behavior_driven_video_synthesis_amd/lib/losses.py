"""MI355X-native counterparts of the hot subset of the reference's ``lib/losses.py`` (:11-14, :26-37,
:55-119, :129-149, :283-291): same function names, argument meaning and return types; the reductions
run in the fused HIP kernels (``vunet_l1_mean_*``, ``vunet_kl_*``, ``vunet_sqdiff_*``).
"""
from __future__ import annotations

from collections import namedtuple

import torch
from torch import nn

from .. import ops
from .utils import get_member

VGGOutput = namedtuple("VGGOutput", ["input", "relu1_2", "relu2_2", "relu3_2", "relu4_2", "relu5_2"])


def _sum_terms(terms):
    total = None
    for t in terms:
        total = t if total is None else total + t
    return total


def latent_kl(prior_mean, posterior_mean):
    """:26-37 -- 0.5 (p - q)^2 summed over CHW, batch mean (one fused reduction)."""
    return ops.SqDiff.apply(prior_mean, posterior_mean, 1.0)


def compute_kl_loss(prior_means, posterior_means):
    """:55-65 -- sum of ``latent_kl`` over the latent scales."""
    return _sum_terms(latent_kl(p, q) for p, q in zip(prior_means, posterior_means))


def aggregate_kl_loss(prior_means, posterior_means):
    """:40-52 -- the dict-valued variant."""
    return compute_kl_loss(list(prior_means.values()), list(posterior_means.values()))


def kl_loss(mu, logstd):
    """:283-291 -- KL(N(mu, exp(logstd)^2) || N(0, 1)) per sample (trailing dims flattened), batch mean."""
    return ops.KLPrior.apply(mu, logstd, 1.0)


def compute_kl_with_prior(means, logstds):
    """:68-78 -- mean over the latent scales of the per-scale KL (the 1/S weight rides in the kernel)."""
    w = 1.0 / len(means)
    return _sum_terms(ops.KLPrior.apply(m, l, w) for m, l in zip(means, logstds))


def vgg_loss(custom_vgg, target, pred, weights=None, target_features=None):
    """lib/losses.py:81-119: ``w_i * mean|t_i - p_i|`` per tap, dict of tensors shaped [1].

    ``target_features``: the dict ``custom_vgg(target)`` if the caller has already computed it (e.g. a fixed target).

    The target pass runs without an autograd graph and the VGG weights are frozen: the reference
    builds both (it never disables ``requires_grad`` on VGG), which changes neither the loss nor the
    gradient reaching the generator.
    """
    fast = (weights is None and ops._grad_passthrough and torch.is_grad_enabled() and pred.requires_grad
            and hasattr(custom_vgg, "loss_terms"))
    if target_features is None:
        with torch.no_grad():
            # (on the tap-by-tap route the target's features may stay in the stack's internal layout)
            target_features = (custom_vgg.features_for_loss(target) if (fast and hasattr(custom_vgg, "features_for_loss"))
                               else custom_vgg(target))
    planes = type(target_features).__name__ == "_P2Target"   # (features_for_loss: the taps in the stack's internal layout)
    if planes and weights is not None:
        raise RuntimeError("vgg_loss(weights=...) needs fp32 target features: pass custom_vgg(target), not features_for_loss(target)")
    if fast or planes:
        return custom_vgg.loss_terms(pred, target_features)   # the same terms, formed tap by tap during the pass
    wanted = VGGOutput(**target_features)
    got = VGGOutput(**custom_vgg(pred))
    tap_weights = get_member(custom_vgg, "loss_weights")
    losses = {name: ops.L1Mean.apply(t, p, float(w))
              for name, t, p, w in zip(VGGOutput._fields, wanted, got, tap_weights)}
    if weights is not None:
        # :103-117 -- a pixel-weight map on the first ("input") tap only: mean(weights * |t0 - p0|).  |w t - w p| =
        # w |t - p| for w >= 0, so the fused L1 kernel is fed the pre-weighted tensors.
        w = weights.to(got[0].dtype).expand_as(got[0])
        losses[VGGOutput._fields[0]] = ops.L1Mean.apply(wanted[0] * w, got[0] * w, float(tap_weights[0]))
    return losses


class GANLoss(nn.Module):
    """lib/losses.py:129-149.  The logits are [N,1]-sized; the scalar loss itself is host-side plumbing."""

    def __init__(self, loss_type: str = "mse"):
        super().__init__()
        if loss_type == "vanilla":
            self.loss = nn.BCEWithLogitsLoss()
        elif loss_type == "mse":
            self.loss = nn.MSELoss()
        else:
            raise ValueError(
                f'The loss type for GANLoss must be either "vanilla" or "mse", but is actually {loss_type}.')
        self.loss_type = loss_type

    def forward(self, pred, target):
        return self.loss(pred, target)
