"""Pure data parallelism for the VUnet training step: one process per GPU, RCCL over xGMI.

Replaces the reference's single-process ``nn.DataParallel`` (experiments/shape_and_pose_net.py:213-214,
223-224, 230-233), whose per-step parameter broadcast, input scatter and output gather disappear:
each rank keeps a persistent replica and its own shard of the batch, and the only exchange is the
gradient average -- an all-reduce per flat parameter bucket (optim.FlatBucket: dd / ed / du / eu),
launched asynchronously as soon as the last gradient of the bucket has been written, so it overlaps
with the rest of the backward pass.  ``backend="nccl"`` is RCCL on ROCm; the same code runs on gloo
for the CPU tests.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist


class BucketedGradAverager:
    def __init__(self, buckets, process_group=None, overlap: bool = True):
        self.buckets = list(buckets)
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # VUNET_DP_FORCE=1: run the full hook / async all-reduce machinery even with one rank (1-GPU smoke test)
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get("VUNET_DP_FORCE") == "1")
        self.overlap = overlap
        self._works = []
        self._pending: List[int] = []
        self._expected: List[Optional[int]] = [None] * len(self.buckets)  # learnt on the first step
        self._fired: List[int] = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._fired_at_launch = [0] * len(self.buckets)
        if self.active:
            from . import ops
            self._bucket_of = {}
            for bi, b in enumerate(self.buckets):
                for p in b.params:
                    self._bucket_of[id(p)] = bi
                    # gradients that autograd accumulates itself ...
                    p.register_post_accumulate_grad_hook(self._make_hook(bi))
            # ... and those the HIP weight-norm backward writes straight into the flat bucket
            ops.add_grad_hook(self._direct_hook)

    def _make_hook(self, bi):
        def hook(param):
            self._fired[bi] += 1
            if self.overlap and self._expected[bi] is not None and self._fired[bi] == self._expected[bi]:
                self._launch(bi)
        return hook

    def _direct_hook(self, param):
        bi = self._bucket_of.get(id(param))
        if bi is not None:
            self._make_hook(bi)(param)

    def _launch(self, bi):
        if self._launched[bi]:
            return
        self._launched[bi] = True
        self._fired_at_launch[bi] = self._fired[bi]
        b = self.buckets[bi]
        if b.grad.is_cuda:
            from . import ops
            ops.join_wgrad_streams()   # in-place gradient writes of backward's companion streams come first
        b.gather_foreign_grads()
        self._works.append(dist.all_reduce(b.grad, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def start_step(self):
        self._fired = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._works = []

    def finish(self, extra_scalars: Optional[torch.Tensor] = None):
        """Flush buckets that were not launched from the hooks, wait for all, scale by 1/world.

        ``extra_scalars`` (e.g. the KL value feeding the gamma controller) is averaged too so every
        rank runs the identical controller (experiments/shape_and_pose_net.py:82-85,442).
        """
        if not self.active:
            return extra_scalars
        for bi in range(len(self.buckets)):
            if self._launched[bi] and self._fired[bi] != self._fired_at_launch[bi]:
                # a gradient landed after the bucket's all-reduce had been launched from the hooks: the firing pattern
                # of backward changed between steps and the reduced values are incomplete -- never continue silently
                raise RuntimeError(f"bucket {self.buckets[bi].name!r}: {self._fired[bi]} gradient writes this step, its "
                                   f"all-reduce was launched after {self._fired_at_launch[bi]} (pattern learnt earlier); "
                                   "construct the averager with overlap=False for graphs that change between steps")
            self._expected[bi] = self._fired[bi]   # learnt on the first step, re-learnt if backward fired fewer hooks
            self._launch(bi)
        if extra_scalars is not None:
            self._works.append(dist.all_reduce(extra_scalars, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        for w in self._works:
            w.wait()
        inv = 1.0 / self.world
        for b in self.buckets:
            b.grad.mul_(inv) if not b.grad.is_cuda else _scale_(b.grad, inv)
        if extra_scalars is not None:
            extra_scalars.mul_(inv)
        return extra_scalars


def _scale_(t: torch.Tensor, a: float):
    from . import ops
    ops._call("vunet_axpby", ops._p(t), None, ops._p(t), float(a), 0.0, t.numel(), ops._stream())


def broadcast_parameters(buckets, src: int = 0, process_group=None):
    """One-time replica initialisation (replaces DataParallel's per-step broadcast)."""
    if dist.is_initialized() and (dist.get_world_size(process_group) > 1 or os.environ.get("VUNET_DP_FORCE") == "1"):
        for b in buckets:
            dist.broadcast(b.flat, src=src, group=process_group)
