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
    written = [key]
    if "regressor" in out:
        # the reference keeps the regressor and its optimiser in a file of their own, ``regressor_*_<it>.pth`` holding
        # {"model", "optimizer"}, and restarts with ``_load_ckpt("regressor")`` (experiments/shape_and_pose_net.py:87-95,
        # 486-497): write that file too, so that ``load_ckpt(dir, "regressor")`` -- and the reference's loader -- find it
        torch.save({"model": out["regressor"]["model"], "optimizer": out["regressor"].get("optimizer")},
                   os.path.join(directory, f"regressor_checkpoint_{int(iteration)}.pth"))
        written.append("regressor")
    for k in written:
        mine = sorted((p for p in glob.glob(os.path.join(directory, f"{k}_checkpoint_*.pth"))),
                      key=lambda p: float(os.path.basename(p).rsplit("_", 1)[-1].split(".")[0]))
        for old in mine[:-n_saved]:
            os.remove(old)
    return path


def restore(directory: str, trainer, key: str = "reg_ckpt") -> bool:
    """Restart ``trainer`` (a ``ShapePoseNet``) from the newest checkpoint of ``directory`` the way the reference does
    (:87-95, :248-255): the ``key`` file for the VUnet + Adam, the ``regressor`` file for the regressor + its Adam
    (falling back to the ``regressor`` entry older files of this package carried inside the ``key`` file)."""
    path = latest_checkpoint(directory, key) if os.path.isdir(directory) else None
    if path is None:
        return False
    ckpt = torch.load(path, map_location="cpu")
    reg_model, reg_opt = load_ckpt(directory, "regressor")
    if reg_model is not None:
        ckpt["regressor"] = {"model": reg_model, "optimizer": reg_opt}
    trainer.load_state_dict(ckpt)
    return True
