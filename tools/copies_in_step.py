#!/usr/bin/env python3
"""Where do the runtime copies (hipMemcpyAsync -> __amd_rocclr_copyBuffer) inside one training step come from?
torch.profiler with stacks, grouped by the innermost frames of this package.   python tools/copies_in_step.py"""
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
cnt = collections.Counter()
for e in prof.events():
    name = e.name
    if name.startswith("aten::") and any(k in name for k in ("copy_", "_to_copy", "to", "clone", "contiguous", "tensor", "fill_", "zero_", "zeros", "cat", "add")):
        if e.cpu_parent is not None and e.cpu_parent.name.startswith("aten::"):
            continue   # count top-level ATen calls only
        frames = [f for f in (e.stack or []) if "behavior_driven" in f or "bench" in f]
        cnt[(name, frames[0].strip() if frames else "?")] += 1
for (name, where), n in cnt.most_common(40):
    print(f"{n:4d}  {name:22s} {where}")
