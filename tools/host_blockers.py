#!/usr/bin/env python3
"""Which runtime calls block the host inside one training step (torch.profiler, CPU side): hipMalloc / hipFree /
synchronisations / copies, by total host time.     python tools/host_blockers.py     (needs an MI355X)"""
import contextlib
import os
import sys
import time

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
for _ in range(4):
    tr.train_fn(batch)
torch.cuda.synchronize()
ts = []
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train_fn(batch)
    ts.append(time.perf_counter() - t0)
torch.cuda.synchronize()
print("host issue ms per step:", [round(1e3 * t, 2) for t in ts], "reserved MB", torch.cuda.memory_reserved() >> 20,
      "allocated MB", torch.cuda.memory_allocated() >> 20, "num_alloc_retries", torch.cuda.memory_stats().get("num_alloc_retries"),
      "segments", torch.cuda.memory_stats().get("segment.all.allocated"))
with profile(activities=[ProfilerActivity.CPU]) as prof:
    torch.cuda.synchronize()
    tr.train_fn(batch)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.key.startswith("hip") or e.key.startswith("cuda")]
rows.sort(key=lambda e: -e.cpu_time_total)
for e in rows[:12]:
    print(f"{e.key:40s} calls {e.count:6d}  host {e.cpu_time_total / 1e3:9.2f} ms")
