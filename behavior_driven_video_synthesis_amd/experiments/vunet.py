"""Training-loop surface of the reference's ``experiments/vunet.py`` (``VunetOrg``) for the MI355X path.

``train_fn`` (:278-338): ``VunetOrg`` forward -> VGG19 perceptual loss (``ll_weight`` x sum of the six tap terms)
+ ``kl_weight`` x ``compute_kl_loss(p_means, q_means)`` -> backward -> Adam(4 param groups); ``lr`` decays
linearly to 0, ``kl_weight`` ramps linearly from ``kl_init`` to ``kl_max`` between 1/2 and 3/4 of the schedule
(:248-266).  Same MI355X-first differences as ``shape_and_pose_net.ShapePoseNet``: fused flat-bucket Adam,
device-resident scalars, bucketed RCCL gradient averaging.  Checkpoint: ``{"model", "optimizer"}``.
"""
from __future__ import annotations

import os

from typing import Dict, Optional

import torch
import torch.distributed as dist

from .. import ops
from ..lib.losses import compute_kl_loss, vgg_loss
from ..lib.utils import get_member, linear_var, n_parameters
from ..models.imagenet_pretrained import PerceptualVGG, vgg19
from ..models.vunets import VunetOrg
from ..optim import FusedAdam
from ..parallel import BucketedGradAverager, broadcast_parameters

DEFAULT_CONFIG = {
    # config/vunet.yaml (DeepFashion)
    "general": {"seed": 42, "debug": False},
    "data": {"dataset": "DeepFashion", "spatial_size": 256, "box_factor": 2, "bottleneck_factor": 2,
             "inplane_normalize": True},
    "architecture": {"n_latent_scales": 2, "conv_layer_type": "l1", "nf_start": 32, "nf_max": 128,
                     "subpixel_upsampling": True, "n_scales": 0, "n_rnb": 2},
    "training": {"batch_size": 8, "vgg_weights": [1.0] * 6, "dropout_prob": 0.05, "lr": 0.0008, "kl_init": 1e-6,
                 "kl_max": 1.0, "n_init_batches": 4, "adam_betas": (0.5, 0.9), "end_iteration": 300000,
                 "ll_weight": 5.0},
}


class Vunet:
    def __init__(self, config: Dict, device="cuda:0", n_channels_x: int = 30, vgg_weights_path: Optional[str] = None,
                 vgg_width_div: int = 1, total_steps: Optional[int] = None, process_group=None,
                 vgg_synthetic: bool = False, vgg_seed: int = 1234):
        self.config = config
        self.device = torch.device(device)
        arch, data, tr = config["architecture"], config["data"], config["training"]
        torch.manual_seed(config["general"].get("seed", 42))
        kw = dict(arch)
        kw.update(data)
        kw["dropout_prob"] = tr.get("dropout_prob", 0.0)
        self.vunet = VunetOrg(n_channels_x=n_channels_x, **kw).to(self.device)
        overlap = self.device.type == "cuda" and os.environ.get("VUNET_TWO_STREAMS", "1") != "0"
        self.vunet.enable_two_streams(overlap)   # pose encoder (du) on a second HIP stream beside eu / ed
        ops.enable_wgrad_streams(overlap)
        self.vgg = vgg19(pretrained=True, weights_path=vgg_weights_path, width_div=vgg_width_div, seed=vgg_seed,
                         synthetic=vgg_synthetic).to(self.device)
        self.vgg.eval()
        self.custom_vgg = PerceptualVGG(self.vgg, tr["vgg_weights"]).to(self.device)
        self.optimizer = FusedAdam(
            [{"params": list(get_member(self.vunet, n).parameters()), "name": n} for n in ("eu", "ed", "du", "dd")],
            lr=tr["lr"], betas=tuple(tr["adam_betas"]))
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        broadcast_parameters(self.optimizer.buckets, 0, process_group)
        self.averager = BucketedGradAverager(self.optimizer.buckets, process_group)
        self.total_steps = total_steps if total_steps is not None else tr["end_iteration"]
        self.iteration = 0
        self.lr = self._adjust_lr(0)
        self.kl_weight = self._adjust_kl_weight(0)
        for pg in self.optimizer.param_groups:
            pg["lr"] = self.lr
        print(f"Number of trainable params is {n_parameters(self.vunet)}")

    def _adjust_lr(self, it):
        lr0 = self.config["training"]["lr"]
        return float(linear_var(it, 0, self.total_steps, lr0, 0, 0, lr0))

    def _adjust_kl_weight(self, it):
        tr = self.config["training"]
        return float(linear_var(it, self.total_steps // 2, 3 * self.total_steps // 4, tr["kl_init"], tr["kl_max"],
                                tr["kl_init"], 1.0))

    def train_fn(self, batch: Dict[str, torch.Tensor], eps=None, prior_eps=None) -> Dict[str, torch.Tensor]:
        tr = self.config["training"]
        if not self.vunet.training:   # (Module.train() walks all ~400 sub-modules: 1.5 ms per step)
            self.vunet.train()
        self.iteration += 1
        self.averager.start_step()
        self.optimizer.zero_grad()
        app_img = batch.get("pose_img_inplane", batch["pose_img"])
        target_img, shape_img = batch["pose_img"], batch["stickman"]
        out_img, q_means, p_means, _ = self.vunet(app_img, shape_img, eps, prior_eps)
        ld = vgg_loss(self.custom_vgg, target_img, out_img)
        likelihood_loss = tr["ll_weight"] * torch.sum(torch.stack([ld[k] for k in ld], dim=0))
        kl_loss = compute_kl_loss(p_means, q_means)
        loss = likelihood_loss + self.kl_weight * kl_loss
        loss.backward()
        self.vunet.join_streams()
        ops.join_wgrad_streams()
        self.averager.finish()
        self.optimizer.step()
        it = self.iteration
        self.lr = self._adjust_lr(it)
        self.kl_weight = self._adjust_kl_weight(it)
        for pg in self.optimizer.param_groups:
            pg["lr"] = self.lr
        out = {"loss": loss.detach(), "likelihood_loss": likelihood_loss.detach(), "kl_loss": kl_loss.detach(),
               "learning_rate": self.lr, "kl_weight": self.kl_weight}
        out.update({k: v.detach() for k, v in ld.items()})
        return out

    def state_dict(self):
        return {"model": self.vunet.state_dict(), "optimizer": self.optimizer.state_dict()}

    def load_state_dict(self, ckpt):
        self.vunet.load_state_dict(ckpt["model"])
        if ckpt.get("optimizer") is not None:
            self.optimizer.load_state_dict(ckpt["optimizer"])
            states = list(ckpt["optimizer"]["state"].values())
            if states:
                self.iteration = int(states[-1]["step"])
