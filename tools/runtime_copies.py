#!/usr/bin/env python3
"""Which host calls put the runtime's own kernels (__amd_rocclr_copyBuffer / fillBuffer: hipMemcpyAsync, hipMemsetAsync) into
one training step?  Device events are matched to the runtime call of the same correlation id, and that call to the innermost
ATen op (and the nearest frame of this package) enclosing it on the host thread.    python tools/runtime_copies.py"""
import collections
import contextlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from bench import make_config, parse  # noqa: E402
from behavior_driven_video_synthesis_amd.experiments.shape_and_pose_net import ShapePoseNet, synthetic_batch  # noqa: E402

sys.argv = sys.argv[:1]
args = parse()
with contextlib.redirect_stdout(sys.stderr):
    tr = ShapePoseNet(make_config(args), device="cuda:0", total_steps=150000, vgg_synthetic=True)
batch = synthetic_batch(16, 256, "cuda:0", seed=42)
for _ in range(3):
    tr.train_fn(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.train_fn(batch)
    torch.cuda.synchronize()

evs = list(prof.events())
dev = [e for e in evs if e.device_type == torch.autograd.DeviceType.CUDA]
cpu = [e for e in evs if e.device_type == torch.autograd.DeviceType.CPU]
print("device events by name:", file=sys.stderr)
names = collections.Counter(e.name[:60] for e in dev)
for n, c in names.most_common(12):
    print(f"  {c:4d}  {n}", file=sys.stderr)

cnt = collections.Counter()
for c in cpu:
    for k in (c.kernels or []):
        if not any(t in k.name for t in ("rocclr", "Memcpy", "Memset")):
            continue
        a = c
        while a is not None and not a.name.startswith("aten::"):
            a = a.cpu_parent
        top = a
        while top is not None and top.cpu_parent is not None and top.cpu_parent.name.startswith("aten::"):
            top = top.cpu_parent
        frames = []
        e = top if top is not None else c
        while e is not None and not frames:
            frames = [f for f in (e.stack or []) if "behavior_driven" in f or "bench" in f]
            e = e.cpu_parent
        cnt[(k.name[:28], c.name[:20], top.name if top is not None else "?", frames[0].strip()[-100:] if frames else "?")] += 1
for (dn, rn, an, where), n in cnt.most_common(40):
    print(f"{n:4d}  {dn:28s} {rn:20s} {an:24s} {where}")
