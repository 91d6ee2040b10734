#!/usr/bin/env python3
"""How long the host needs to ISSUE one training step (no device sync) vs how long the GPU needs to run it."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import make_config, parse  # noqa: E402
from behavior_driven_video_synthesis_amd.experiments.shape_and_pose_net import ShapePoseNet, synthetic_batch  # noqa: E402

args = parse()
cfg = make_config(args)
tr = ShapePoseNet(cfg, device="cuda:0", total_steps=150000, vgg_synthetic=True)
batch = synthetic_batch(args.batch, args.size, "cuda:0", seed=42)
for _ in range(3):
    tr.train_fn(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    tr.train_fn(batch)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host issue {1e3 * t_issue / args.steps:.2f} ms/step, wall {1e3 * t_all / args.steps:.2f} ms/step")
import cProfile
import pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    tr.train_fn(batch)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
