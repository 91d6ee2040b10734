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


def synth_behavior_state(shapes: Dict[str, Sequence[int]], seed: int, stored: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """State dict of a flow / behaviour net: ``stored`` holds the tensors the recipe cannot rebuild (the ``Shuffle``
    permutations the reference drew); ActNorm is marked initialised, with scales away from zero and of both signs;
    recurrent weights are scaled by fan-in."""
    out = {}
    for k, shape in shapes.items():
        leaf = k.rsplit(".", 1)[-1]
        if k in stored:
            out[k] = torch.as_tensor(stored[k]).clone()
        elif leaf == "initialized":
            out[k] = torch.tensor(1, dtype=torch.uint8)
        elif leaf == "scale":
            r = seeded_randn(k, tuple(shape), seed)
            sign = torch.where(r > 1.2, -torch.ones_like(r), torch.ones_like(r))
            out[k] = (0.7 + 0.3 * r.abs()) * sign
        elif leaf in ("weight_ih", "weight_hh", "weight_ih_l0", "weight_hh_l0"):
            out[k] = seeded_randn(k, tuple(shape), seed) * (1.0 / shape[1]) ** 0.5
        else:
            out[k] = synth_param(k, tuple(shape), seed)
    return out
