"""The two training stages of ``experiments/behavior_net.py`` (BASELINE config 4) on the MI355X path.

``BehaviorNet.train_fn(batch)`` is the reference's ``train_fn`` (:446-760) for a batch ``{"keypoints": [B, T + 1, n_kps]}``:

* first stage (``only_flow`` False, :591-660): ``net(seq_b, seq_b, seq_len)`` -> ``recon_loss_weight * mean(MSE) + gamma *
  kl_loss`` -> backward -> ``Adam(lr_init)`` -> the gamma controller (:111-116).  Forward, loss, back-propagation through
  time and the fused optimiser step are one recorded hipGraph (csrc/seq.hip, csrc/seq_bptt.hip, csrc/seq_train.hip).
* flow stage (``only_flow`` True, :703-714): the net encodes under ``no_grad``, then ``latent_flow(bs.detach())`` ->
  ``FlowLoss`` -> backward -> ``Adam(flow_lr * batch_size, betas (0.5, 0.9), weight_decay)`` with the weight gradient fused
  into the update (``seq_train.FlowTrainEngine``).

Not on this path, and said here rather than silently dropped:
* the second pass ``net(seq_2, seq_start_t, seq_len)`` (:600-603): none of its outputs reaches the loss or the log;
* ``use_regressor`` (:627-648): the reference steps the regressor's weights in place between the loss's forward and its
  backward, which torch >= 1.5 refuses (tests/golden/g11_cvae_training.npz records the error); the trainer raises if asked;
* the action classifiers (:662-690): models/pose_discriminator.py is outside the path and they feed no
  gradient that the net's optimiser uses.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import torch

from ..models.flow.simple_flow import UnsupervisedTransformer2
from ..models.pose_behavior_rnn import ResidualBehaviorNet
from ..ops import _call, _p, _stream
from ..optim import FusedAdam

DEFAULT_CONFIG = {   # config/behavior_net.yaml
    "architecture": {"decoder_arch": "lstm", "linear_in_decoder": False, "dim_hidden_b": 1024, "flow_mid_channels_factor": 2,
                     "n_flows": 15, "flow_hidden_depth": 2, "cvae": False},
    "training": {"batch_size": 64, "n_epochs": 50, "lr_init": 1e-4, "tau": [0.2, 0.45, 0.7], "gamma": 0.3, "weight_decay": 0.0,
                 "recon_loss_weight": 2.5, "information_max": 100, "gamma_init": 0, "gamma_step": 1e-5, "flow_lr": 4.5e-7,
                 "use_regressor": False, "weight_regressor": 0.01, "imax_scaling": "none", "only_flow": False},
}


class BehaviorNet:
    def __init__(self, config: Optional[dict] = None, n_kps: int = 51, device="cuda", hip_graph: bool = True):
        import copy
        self.config = copy.deepcopy(DEFAULT_CONFIG if config is None else config)
        arch, tr = self.config["architecture"], self.config["training"]
        if tr.get("use_regressor", False):
            raise NotImplementedError("use_regressor: the reference's own step raises on torch >= 1.5 (an in-place optimiser step "
                                      "between the loss's forward and backward, experiments/behavior_net.py:636-653); not built")
        self.device = torch.device(device)
        self.only_flow = bool(tr.get("only_flow", False))
        self.net = ResidualBehaviorNet(n_kps=n_kps, information_bottleneck=True, **arch).to(self.device)     # :310-313
        self.optimizer = FusedAdam([{"params": list(self.net.b_enc.parameters()), "name": "z_enc"},
                                    {"params": list(self.net.decoder.parameters()), "name": "dec"}], lr=tr["lr_init"])   # :324-331
        self.optimizer.use_device_schedule()
        self.latent_flow = UnsupervisedTransformer2(flow_in_channels=arch["dim_hidden_b"], n_flows=arch["n_flows"],
                                                    flow_hidden_depth=arch["flow_hidden_depth"],
                                                    flow_mid_channels=arch["dim_hidden_b"] * arch["flow_mid_channels_factor"]).to(self.device)
        self.flow_lr = tr["flow_lr"] * tr["batch_size"]                                                          # :382
        self.flow_engine = self.latent_flow.flow.train_engine(lr=self.flow_lr, betas=(0.5, 0.9), weight_decay=tr["weight_decay"])
        self.gamma_dev = torch.full((1,), float(tr["gamma_init"]), device=self.device)
        self.imax = float(tr["information_max"])
        self.imax_dev = torch.full((1,), self.imax, device=self.device)
        self.hip_graph = hip_graph

    # ---- the reference's knobs
    @property
    def gamma(self) -> float:
        return float(self.gamma_dev.item())

    def set_imax(self, imax: float):
        if float(imax) != self.imax:
            self.imax = float(imax)
            self.imax_dev.fill_(self.imax)

    def set_lr(self, lr: float):
        """``MultiStepLR`` (:337-339) steps the net's learning rate per epoch."""
        for g in self.optimizer.param_groups:
            g["lr"] = float(lr)
        self.optimizer.lr_dev.fill_(float(lr))
        self.optimizer._lr_mirrored = float(lr)

    # ---- one step
    def _cvae_step(self, seq_b, target, eps) -> Dict:
        tr = self.config["training"]
        eng = self.net.train_engine()
        eng._check()
        rows, t_in = seq_b.shape[0], seq_b.shape[1]
        p = eng._tplan(rows, t_in, t_in, t_in)
        p["x1"].copy_(seq_b)
        p["x2"].copy_(seq_b)
        p["target"].copy_(target)
        if eps is None:
            torch.randn(p["eps"].shape, out=p["eps"])
        else:
            p["eps"].copy_(eps)
        tuning_is_gamma = not self.config["architecture"].get("cvae", False)
        if not tuning_is_gamma:
            raise NotImplementedError("cvae: True (a fixed KL weight of 1) is not wired; config/behavior_net.yaml trains with False")
        grads = {n: q.grad for n, q in self.net.named_parameters()}

        def issue():
            eng._fill_images()
            eng._issue_train_forward(rows, p, t_in, t_in, t_in, 0, False)
            _call("vunet_seq_vae_loss", _p(p["xs"]), _p(p["target"]), _p(p["mu"]), _p(p["logstd"]), rows, t_in, eng.n, eng.H,
                  float(tr["recon_loss_weight"]), _p(self.gamma_dev), _p(self.imax_dev), float(tr["gamma_step"]), _p(p["part"]),
                  _p(p["scalars"]), _p(p["per_seq"]), _p(p["gxs"]), _p(p["gmu"]), _p(p["glogstd"]), _stream())
            eng._issue_train_backward(rows, p, t_in, t_in, False)
            eng._unpack_grads(p, grads)
            self.optimizer.step()
        eng.graph.enabled = self.hip_graph
        mode = eng.graph.run_step(("cvae", rows, t_in), issue)
        if mode == "replayed":
            self.optimizer.note_replayed_steps(1)
        eng._packed_for = None     # the parameters moved: an inference call re-fills the images
        return p

    def train_fn(self, batch: Dict[str, torch.Tensor], eps: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None,
                 sync: bool = True) -> Dict:
        """``eps``: the reparametrisation noise (None: drawn); ``noise``: the draw behind the flow stage's logged
        ``reference_nll_loss``.  ``sync`` False returns device tensors instead of floats (no host round trip per step)."""
        kps = batch["keypoints"].to(self.device, torch.float32)
        seq_b, target = kps[:, :-1].contiguous(), kps[:, 1:].contiguous()       # prepare_input (lib/utils.py:914-917)
        seq_len = seq_b.shape[1]
        out: Dict = {}
        if not self.only_flow:
            p = self._cvae_step(seq_b, target, eps)
            sc = p["scalars"]
            out.update(loss=sc[0], loss_recon=sc[1], kl_loss=sc[2], gamma=self.gamma_dev[0], mu_s=sc[4], logstd_s=sc[5],
                       loss_per_seq_recon=p["per_seq"])
        else:
            with torch.no_grad():
                xs, cs, _, bs, mu_s, logstd_s, pre_s = self.net(seq_b, seq_b, seq_len, eps=eps)
                eng = self.net.train_engine()
                eng._check()
                p = eng._tplan(seq_b.shape[0], seq_len, seq_len, seq_len)
                _call("vunet_seq_vae_loss", _p(xs), _p(target), _p(mu_s), _p(logstd_s), seq_b.shape[0], seq_len, eng.n, eng.H,
                      float(self.config["training"]["recon_loss_weight"]), _p(self.gamma_dev), _p(self.imax_dev), 0.0, _p(p["part"]),
                      _p(p["scalars"]), _p(p["per_seq"]), None, None, None, _stream())
                sc = p["scalars"].clone()
            self.flow_engine.graph.enabled = self.hip_graph
            fl = self.flow_engine.train_step(bs.detach(), noise)
            out.update(flow_loss=fl[0], reference_nll_loss=fl[1], nlogdet_loss=fl[2], nll_loss=fl[3], loss_recon=sc[1], kl_loss=sc[2],
                       gamma=self.gamma_dev[0], mu_s=sc[4], logstd_s=sc[5], loss_per_seq_recon=p["per_seq"])
        out["imax"], out["seq_len"] = self.imax, seq_len
        if sync:
            out = {k: (v.detach().cpu().numpy() if k == "loss_per_seq_recon" else float(v)) if isinstance(v, torch.Tensor) else v
                   for k, v in out.items()}
        return out

    # ---- checkpoints in the reference's layout (:1003-1011: model / optimizer / flow / flow optimizer)
    def state_dict(self) -> dict:
        return {"model": self.net.state_dict(), "optimizer": self.optimizer.state_dict(), "flow": self.latent_flow.state_dict(),
                "flow_optimizer": self.flow_engine.optimizer_state_dict(), "gamma": self.gamma}

    def load_state_dict(self, sd: dict):
        self.net.load_state_dict(sd["model"])
        if sd.get("optimizer") is not None:
            self.optimizer.load_state_dict(sd["optimizer"])
        if sd.get("flow") is not None:
            self.latent_flow.load_state_dict(sd["flow"])
        if sd.get("flow_optimizer") is not None:
            self.flow_engine.load_optimizer_state_dict(sd["flow_optimizer"])
        if "gamma" in sd:
            self.gamma_dev.fill_(float(sd["gamma"]))
