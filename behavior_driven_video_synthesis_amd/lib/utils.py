"""Hot-path helpers of the reference's ``lib/utils.py`` (:33-48, :520-527, :658-668)."""
from __future__ import annotations

import numpy as np
from torch import nn


def n_parameters(model):
    """lib/utils.py:33-34."""
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def get_member(model, name):
    """lib/utils.py:37-43: attribute access through an optional nn.DataParallel wrapper."""
    module = model.module if isinstance(model, (nn.DataParallel, nn.parallel.DistributedDataParallel)) else model
    return getattr(module, name)


def toggle_grad(model, requires_grad):
    """lib/utils.py:46-48."""
    for p in model.parameters():
        p.requires_grad_(requires_grad)


def linear_var(act_it, start_it, end_it, start_val, end_val, clip_min, clip_max):
    """lib/utils.py:520-527."""
    act_val = float(end_val - start_val) / (end_it - start_it) * (act_it - start_it) + start_val
    return np.clip(act_val, a_min=clip_min, a_max=clip_max)


def scale_img(x):
    """lib/utils.py:658-668: [-1,1] -> [0,1]."""
    return (x + 1.0) / 2.0
