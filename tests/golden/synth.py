"""Seeded synthetic tensors shared by the golden-vector generator and the tests.

Every tensor is drawn from its own ``torch.Generator`` whose seed is derived
from (seed, name), so values do not depend on creation order and a state dict
can be rebuilt from its (name, shape) list alone -- fixtures then only need to
store shapes, seeds and expected outputs.
"""
from __future__ import annotations

import hashlib
from typing import Dict, Sequence

import torch


def _seed_for(name: str, seed: int) -> int:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return int.from_bytes(h[:7], "little")


def seeded_randn(name: str, shape: Sequence[int], seed: int = 0) -> torch.Tensor:
    g = torch.Generator().manual_seed(_seed_for(name, seed))
    return torch.randn(*shape, generator=g, dtype=torch.float32)


def seeded_rand(name: str, shape: Sequence[int], seed: int = 0) -> torch.Tensor:
    g = torch.Generator().manual_seed(_seed_for(name, seed))
    return torch.rand(*shape, generator=g, dtype=torch.float32)


def synth_param(name: str, shape: Sequence[int], seed: int = 0) -> torch.Tensor:
    """Non-trivial values for every parameter kind of the hot path."""
    leaf = name.rsplit(".", 1)[-1]
    r = seeded_randn(name, shape, seed)
    if leaf == "gamma":
        return 1.0 + 0.3 * r
    if leaf == "beta":
        return 0.2 * r
    if leaf == "weight_g":
        return 0.25 * (r.abs() + 0.5)
    if leaf == "weight_v":
        return 0.2 * r
    if leaf == "bias":
        return 0.1 * r
    if leaf == "weight":
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        return r * (1.5 / max(fan_in, 1)) ** 0.5
    return 0.1 * r


def synth_state_dict(shapes: Dict[str, Sequence[int]], seed: int = 0) -> Dict[str, torch.Tensor]:
    return {k: synth_param(k, tuple(v), seed) for k, v in shapes.items()}


def synth_image(name: str, shape: Sequence[int], seed: int = 0) -> torch.Tensor:
    """Image-like input in [-1, 1]."""
    return seeded_rand(name, shape, seed) * 2.0 - 1.0
