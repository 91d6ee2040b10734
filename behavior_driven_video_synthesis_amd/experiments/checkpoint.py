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


def load_pretrained(pretrained_model: str, device="cuda:0", run_dir: Optional[str] = None, key: str = "reg_ckpt", **trainer_kw):
    """The reference's ``--pretrained_model <dir>`` flow (main.py:38-47, experiments/experiment.py:39-95,
    experiments/shape_and_pose_net.py:87-95, 248-255): read ``<dir>/config.yaml`` (the YAML the authors ship beside
    their checkpoints; ``!!python/tuple`` tags as the reference's FullLoader reads them), build the trainer that
    config describes, and restore -- strictly -- the newest ``*.pth`` of the directory whose name contains ``key``:
    ``{"model": VunetAlter.state_dict(), "optimizer": torch.optim.Adam.state_dict()}``, Adam's per-parameter state
    included (iteration, gamma, lr / imax schedules follow from it as on a restart).  ``run_dir``: also do what
    main.py does with the directory -- copy the config to ``run_dir/config/config.yaml`` and every ``*.pth`` to
    ``run_dir/ckpt``.  -> (trainer, config dict).  ``trainer_kw`` goes to ``ShapePoseNet`` (e.g. ``vgg_weights_path``)."""
    import shutil
    import yaml
    from .shape_and_pose_net import ShapePoseNet
    cfg_path = os.path.join(pretrained_model, "config.yaml")
    if not os.path.isfile(cfg_path):
        raise FileNotFoundError("No saved config file found but model is intended to be restarted. Aborting....")
    with open(cfg_path, "r") as f:
        cdict = yaml.load(f, Loader=yaml.FullLoader)
    path = latest_checkpoint(pretrained_model, key)
    if path is None:
        raise FileNotFoundError(f"no *{key}*.pth in {pretrained_model}")
    if run_dir is not None:
        for sub in ("config", "ckpt"):
            os.makedirs(os.path.join(run_dir, sub), exist_ok=True)
        with open(os.path.join(run_dir, "config", "config.yaml"), "w") as f:
            yaml.dump(cdict, f, default_flow_style=False)
        for c in glob.glob(os.path.join(pretrained_model, "*.pth")):
            shutil.copy(c, os.path.join(run_dir, "ckpt"))
    config = {k: dict(v) if isinstance(v, dict) else v for k, v in cdict.items()}
    config.setdefault("general", {}).setdefault("seed", 42)
    trainer = ShapePoseNet(config, device=device, **trainer_kw)
    ckpt = torch.load(path, map_location="cpu")
    if "model" not in ckpt:
        raise KeyError(f"{path}: no 'model' entry (the reference's checkpoint layout is {{'model', 'optimizer'}})")
    reg_model, reg_opt = load_ckpt(pretrained_model, "regressor")
    if reg_model is not None:
        ckpt["regressor"] = {"model": reg_model, "optimizer": reg_opt}
    trainer.load_state_dict(ckpt)   # strict for the VUnet; Adam state, iteration, gamma, schedules as on a restart
    return trainer, config

