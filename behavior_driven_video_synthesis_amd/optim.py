"""Fused Adam over flat fp32 parameter buckets (vunet_adam_step), torch.optim.Adam-compatible surface.

Replaces ``torch.optim.Adam([...4 param groups...], lr, betas)`` of
experiments/shape_and_pose_net.py:237-246: each param group is flattened once into one contiguous
parameter buffer and one contiguous gradient buffer (the parameters and their ``.grad`` become views),
so that the optimiser step is one HBM-bound launch per group and the gradient all-reduce of the
data-parallel path (parallel.py) works on the same flat buckets without packing copies.
``state_dict``/``load_state_dict`` use torch.optim.Adam's layout (per-parameter ``step / exp_avg /
exp_avg_sq``; extra group keys such as ``name`` and ``gamma`` are preserved, :507-512).
"""
from __future__ import annotations

from typing import Dict, List

import torch

from . import ops


class FlatBucket:
    """One contiguous fp32 buffer holding a list of parameters (and one holding their gradients)."""

    def __init__(self, params: List[torch.nn.Parameter], name: str = ""):
        self.name = name
        self.params = [p for p in params]
        assert self.params, "empty parameter group"
        dev = self.params[0].device
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.empty(self.numel, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        self.offsets = []
        off = 0
        with torch.no_grad():
            for p in self.params:
                n = p.numel()
                self.flat[off:off + n].copy_(p.detach().reshape(-1))
                p.data = self.flat[off:off + n].view(p.shape)
                p.grad = self.grad[off:off + n].view(p.shape)
                p._vunet_direct_grad = True  # ops.FusedConv writes this view in place (no autograd add pass)
                self.offsets.append(off)
                off += n

    def zero_grad(self):
        self.grad.zero_()
        for p, off in zip(self.params, self.offsets):  # re-attach in case autograd replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                p.grad = self.grad[off:off + p.numel()].view(p.shape)

    def gather_foreign_grads(self):
        """If autograd swapped a .grad tensor for its own (it does when .grad was None), copy it back."""
        for p, off in zip(self.params, self.offsets):
            if p.grad is not None and p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                self.grad[off:off + p.numel()].copy_(p.grad.reshape(-1))
                p.grad = self.grad[off:off + p.numel()].view(p.shape)


class FusedAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        groups = list(params)
        if groups and not isinstance(groups[0], dict):
            groups = [{"params": groups}]
        self.defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False)
        self.param_groups: List[Dict] = []
        self.buckets: List[FlatBucket] = []
        for g in groups:
            g = dict(g)
            g["params"] = list(g["params"])
            for k, v in self.defaults.items():
                g.setdefault(k, v)
            b = FlatBucket(g["params"], g.get("name", ""))
            b.exp_avg = torch.zeros_like(b.flat)
            b.exp_avg_sq = torch.zeros_like(b.flat)
            b.step = 0
            self.buckets.append(b)
            self.param_groups.append(g)
        self.lr_dev = self.step_dev = None

    def use_device_schedule(self, lr_dev: torch.Tensor = None):
        """Keep the step count (int64) and the learning rate (float64, ``lr_dev`` or a tensor of its own) on the device:
        ``step()`` then issues no launch argument that changes between steps (vunet_adam_step_dev) and can be captured
        in a hipGraph.  The caller writes ``lr_dev`` (outside the graph); the step count advances inside ``step()``.
        One learning rate for all groups -- what the reference's schedule sets (:500-512)."""
        dev = self.buckets[0].flat.device
        self._lr_dev_own = lr_dev is None      # nobody else writes it: step() mirrors param_groups[*]["lr"] into it
        self._lr_mirrored = float(self.param_groups[0]["lr"])
        self.lr_dev = lr_dev if lr_dev is not None else torch.full((1,), self._lr_mirrored, dtype=torch.float64,
                                                                   device=dev)
        self.step_dev = torch.full((1,), int(self.buckets[0].step), dtype=torch.int64, device=dev)
        return self

    def note_replayed_steps(self, n: int = 1):
        """A captured graph containing ``n`` calls of ``step()`` was replayed: advance the host-side step counts (the
        checkpoint layout carries them) without launching anything."""
        for b in self.buckets:
            b.step += n

    def zero_grad(self, set_to_none: bool = False):
        for b in self.buckets:
            b.zero_grad()

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0):
        if self.buckets[0].grad.is_cuda:
            ops.join_wgrad_streams()   # gradients written in place by backward's companion streams come first
        if self.step_dev is not None:
            lrs = {float(g["lr"]) for g in self.param_groups}
            if len(lrs) != 1:
                raise RuntimeError("device-schedule mode keeps ONE learning rate for all param groups "
                                   f"(vunet_adam_step_dev); the groups carry {sorted(lrs)}")
            if self._lr_dev_own and not torch.cuda.is_current_stream_capturing():
                lr = lrs.pop()
                if lr != self._lr_mirrored:   # a later pg["lr"] change / load_state_dict: one fill outside any capture
                    self.lr_dev.fill_(lr)
                    self._lr_mirrored = lr
            self.step_dev.add_(1)
            for g, b in zip(self.param_groups, self.buckets):
                b.gather_foreign_grads()
                b.step += 1
                ops.adam_step_flat_dev(b.flat, b.grad, b.exp_avg, b.exp_avg_sq, self.lr_dev, g["betas"][0], g["betas"][1],
                                       g["eps"], g["weight_decay"], self.step_dev, grad_scale)
            return
        for g, b in zip(self.param_groups, self.buckets):
            b.gather_foreign_grads()
            b.step += 1
            ops.adam_step_flat(b.flat, b.grad, b.exp_avg, b.exp_avg_sq, g["lr"], g["betas"][0], g["betas"][1],
                               g["eps"], g["weight_decay"], b.step, grad_scale)

    # ---- torch.optim.Adam checkpoint layout
    def state_dict(self):
        state, groups, idx = {}, [], 0
        for g, b in zip(self.param_groups, self.buckets):
            ids = []
            for p, off in zip(b.params, b.offsets):
                n = p.numel()
                if b.step > 0:
                    state[idx] = {"step": torch.tensor(float(b.step)),
                                  "exp_avg": b.exp_avg[off:off + n].view(p.shape).clone(),
                                  "exp_avg_sq": b.exp_avg_sq[off:off + n].view(p.shape).clone()}
                ids.append(idx)
                idx += 1
            groups.append({**{k: v for k, v in g.items() if k != "params"}, "params": ids})
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        idx = 0
        for g, b, sg in zip(self.param_groups, self.buckets, sd["param_groups"]):
            for k, v in sg.items():
                if k != "params":
                    g[k] = v
            steps = []
            for p, off in zip(b.params, b.offsets):
                st = sd["state"].get(idx)
                if st is not None:
                    n = p.numel()
                    b.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
                    b.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                    steps.append(int(st["step"]))
                idx += 1
            b.step = max(steps) if steps else 0
        if self.step_dev is not None:
            self.step_dev.fill_(int(self.buckets[0].step))
