"""Checkpoint files in the reference's layout (experiments/experiment.py:39-95, experiments/shape_and_pose_net.py:471-482).

The reference saves ``{"model": state_dict, "optimizer": adam_state_dict}`` through ignite's
``ModelCheckpoint(dir, "reg_ckpt", n_saved=10)`` -- files named ``<prefix>_<name>_<iteration>.pth`` -- and on restart
loads, among the ``*.pth`` files whose name contains the key, the one whose trailing ``_<number>.pth`` is largest.
"""
from __future__ import annotations

import glob
import os
from typing import Optional, Tuple

import torch


def latest_checkpoint(directory: str, key: str) -> Optional[str]:
    """Path of the newest ``*.pth`` whose file name contains ``key`` (largest trailing ``_<number>``), or None."""
    best, best_it = None, None
    for path in glob.glob(os.path.join(directory, "*.pth")):
        base = os.path.basename(path)
        if key not in base:
            continue
        try:
            it = float(base.rsplit("_", 1)[-1].split(".")[0])
        except ValueError:
            continue
        if best_it is None or it > best_it:
            best, best_it = path, it
    return best


def load_ckpt(directory: str, key: str, only_model: bool = False) -> Tuple[Optional[dict], Optional[dict]]:
    """-> (model state dict, optimizer state dict) like ``Experiment._load_ckpt``; (None, None) if nothing matches."""
    path = latest_checkpoint(directory, key) if os.path.isdir(directory) else None
    if path is None:
        return None, None
    ckpt = torch.load(path, map_location="cpu")
    if only_model:
        return ckpt, None
    return ckpt.get("model"), ckpt.get("optimizer")


def save_ckpt(directory: str, key: str, iteration: int, trainer, n_saved: int = 10) -> str:
    """Write ``trainer.state_dict()`` as ``<key>_checkpoint_<iteration>.pth`` and keep the newest ``n_saved`` files."""
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, f"{key}_checkpoint_{int(iteration)}.pth")
    sd = trainer.state_dict()

    def host(o):
        if isinstance(o, torch.Tensor):
            return o.detach().cpu()
        if isinstance(o, dict):
            return {k: host(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return type(o)(host(v) for v in o)
        return o
    out = {"model": host(sd["model"]), "optimizer": host(sd["optimizer"])}
    for extra in ("regressor", "discriminator"):   # the reference's second checkpoint file / DiscTrainer's own dict
        if extra in sd:
            out[extra] = host(sd[extra])
    torch.save(out, path)
    mine = sorted((p for p in glob.glob(os.path.join(directory, f"{key}_checkpoint_*.pth"))),
                  key=lambda p: float(os.path.basename(p).rsplit("_", 1)[-1].split(".")[0]))
    for old in mine[:-n_saved]:
        os.remove(old)
    return path
