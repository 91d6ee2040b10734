#!/usr/bin/env python3
"""Host time inside the C-ABI calls of one bs-16 training step (ops._call wrapped), by entry point: how much of the
step's issue time is the library + HIP runtime (launch calls) and how much is Python around them."""
import collections
import contextlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import make_config, parse  # noqa: E402
from behavior_driven_video_synthesis_amd import ops  # noqa: E402
from behavior_driven_video_synthesis_amd.experiments.shape_and_pose_net import ShapePoseNet, synthetic_batch  # noqa: E402

sys.argv = sys.argv[:1]
args = parse()
with contextlib.redirect_stdout(sys.stderr):
    tr = ShapePoseNet(make_config(args), device="cuda:0", total_steps=150000, vgg_synthetic=True)
batch = synthetic_batch(16, 256, "cuda:0", seed=42)
for _ in range(5):
    tr.train_fn(batch)
torch.cuda.synchronize()
acc = collections.defaultdict(lambda: [0, 0.0])
orig = ops._call


def timed(name, *a):
    t0 = time.perf_counter()
    try:
        return orig(name, *a)
    finally:
        r = acc[name]
        r[0] += 1
        r[1] += time.perf_counter() - t0


ops._call = timed
steps = 5
tot = 0.0
for _ in range(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train_fn(batch)
    tot += time.perf_counter() - t0
torch.cuda.synchronize()
ops._call = orig
inside = sum(v[1] for v in acc.values())
print(f"issue {1e3 * tot / steps:.2f} ms/step; inside C-ABI calls {1e3 * inside / steps:.2f} ms/step over {sum(v[0] for v in acc.values()) / steps:.0f} calls")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"  {k:34s} {v[0] / steps:6.1f} calls  {1e6 * v[1] / v[0]:6.1f} us each  {1e3 * v[1] / steps:6.2f} ms/step")
