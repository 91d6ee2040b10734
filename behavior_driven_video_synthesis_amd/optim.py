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

    def __init__(self, params: List[torch.nn.Parameter], name: str = "", flat: torch.Tensor = None,
                 grad: torch.Tensor = None):
        """``flat`` / ``grad``: slices of an arena shared by all buckets of an optimiser (FusedAdam), so that zeroing the
        gradients and the Adam update are ONE launch each over the arena; None: buffers of the bucket's own."""
        self.name = name
        self.params = [p for p in params]
        assert self.params, "empty parameter group"
        dev = self.params[0].device
        self.numel = sum(p.numel() for p in self.params)
        self.flat = flat if flat is not None else torch.empty(self.numel, device=dev, dtype=torch.float32)
        self.grad = grad if grad is not None else torch.zeros(self.numel, device=dev, dtype=torch.float32)
        assert self.flat.numel() == self.numel and self.grad.numel() == self.numel
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

    def zero_grad(self, fill: bool = True):
        if fill:
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
        groups = [dict(g) for g in groups]
        for g in groups:
            g["params"] = list(g["params"])
            for k, v in self.defaults.items():
                g.setdefault(k, v)
        # ONE arena for the parameters, gradients and moments of all groups (each bucket a 256-byte-aligned slice): zeroing
        # the gradients and -- while the groups share their hyper-parameters, as the reference's do (:237-246, one lr
        # schedule for all four) -- the Adam update are one launch each instead of one per group.  The padding between
        # slices holds zeros and stays zero (g = 0, m = v = 0: the update is 0).
        dev = groups[0]["params"][0].device if groups and groups[0]["params"] else None
        sizes = [sum(p.numel() for p in g["params"]) for g in groups]
        starts, total = [], 0
        for n in sizes:
            starts.append(total)
            total += (n + 63) // 64 * 64
        self._arena = None
        if dev is not None and len(groups) > 1:
            self._arena = {k: torch.zeros(total, device=dev, dtype=torch.float32) for k in ("flat", "grad", "m", "v")}
        for g, n, st in zip(groups, sizes, starts):
            if self._arena is not None:
                b = FlatBucket(g["params"], g.get("name", ""), self._arena["flat"][st:st + n], self._arena["grad"][st:st + n])
                b.exp_avg, b.exp_avg_sq = self._arena["m"][st:st + n], self._arena["v"][st:st + n]
            else:
                b = FlatBucket(g["params"], g.get("name", ""))
                b.exp_avg = torch.zeros_like(b.flat)
                b.exp_avg_sq = torch.zeros_like(b.flat)
            b.step = 0
            self.buckets.append(b)
            self.param_groups.append(g)
        self.lr_dev = self.step_dev = None

    def _uniform(self):
        """The hyper-parameters all groups share, or None (then every bucket gets its own launch)."""
        if self._arena is None:
            return None
        keys = ("lr", "betas", "eps", "weight_decay")
        first = tuple(self.param_groups[0][k] for k in keys)
        if all(tuple(g[k] for k in keys) == first for g in self.param_groups) and len({b.step for b in self.buckets}) == 1:
            return first
        return None

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
        if self._arena is not None:
            self._arena["grad"].zero_()
        for b in self.buckets:
            b.zero_grad(fill=self._arena is None)

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
            uni = self._uniform()
            for b in self.buckets:
                b.gather_foreign_grads()
                b.step += 1
            if uni is not None:
                ar = self._arena
                ops.adam_step_flat_dev(ar["flat"], ar["grad"], ar["m"], ar["v"], self.lr_dev, uni[1][0], uni[1][1], uni[2],
                                       uni[3], self.step_dev, grad_scale)
                return
            for g, b in zip(self.param_groups, self.buckets):
                ops.adam_step_flat_dev(b.flat, b.grad, b.exp_avg, b.exp_avg_sq, self.lr_dev, g["betas"][0], g["betas"][1],
                                       g["eps"], g["weight_decay"], self.step_dev, grad_scale)
            return
        uni = self._uniform()
        for b in self.buckets:
            b.gather_foreign_grads()
            b.step += 1
        if uni is not None:
            ar = self._arena
            ops.adam_step_flat(ar["flat"], ar["grad"], ar["m"], ar["v"], uni[0], uni[1][0], uni[1][1], uni[2], uni[3],
                               self.buckets[0].step, grad_scale)
            return
        for g, b in zip(self.param_groups, self.buckets):
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
