#!/usr/bin/env python3
"""Host-side time of one bs-16 training step by event (torch.profiler, CPU activity): autograd nodes, ATen ops, runtime
launch calls -- what the 15-24 ms the host needs to ISSUE a step are made of.     python tools/host_breakdown.py"""
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
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train_fn(batch)
    ts.append(1e3 * (time.perf_counter() - t0))
torch.cuda.synchronize()
print("host issue ms per step (unprofiled):", [round(t, 2) for t in ts])
with profile(activities=[ProfilerActivity.CPU]) as prof:
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train_fn(batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
print("profiled step host ms:", round(1e3 * (t1 - t0), 2))
rows = sorted(prof.key_averages(), key=lambda e: -e.self_cpu_time_total)
for e in rows[:30]:
    print(f"{e.key[:60]:60s} n {e.count:5d}  self {e.self_cpu_time_total / 1e3:8.2f} ms  total {e.cpu_time_total / 1e3:8.2f} ms")
