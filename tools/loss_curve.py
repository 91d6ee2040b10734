#!/usr/bin/env python3
"""Loss / KL / gamma every 25 steps of the bench workload (one repeated bs-16 batch) for one conv scheme (needs an MI355X):

    python tools/loss_curve.py h2|x6|f32 [steps]

The first steps agree to 4-5 digits across schemes; later the trajectories separate chaotically (the KL term of a single
repeated batch spikes at different steps in every scheme, fp32 included) -- not a precision effect."""
import sys, copy, contextlib, torch
sys.path.insert(0, "/root/repo")
from behavior_driven_video_synthesis_amd import ops
from behavior_driven_video_synthesis_amd.experiments.shape_and_pose_net import DEFAULT_CONFIG, ShapePoseNet, synthetic_batch
mode = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
ops.set_conv_precision(mode)
cfg = copy.deepcopy(DEFAULT_CONFIG)
with contextlib.redirect_stdout(sys.stderr):
    tr = ShapePoseNet(cfg, device="cuda:0", total_steps=150000, vgg_synthetic=True)
batch = synthetic_batch(16, 256, "cuda:0", seed=42)
out = []
for i in range(steps):
    o = tr.train_fn(batch)
    if i % 25 == 24 or i < 3:
        out.append((i + 1, round(float(o["loss"]), 3), round(float(o["kl_loss"]), 3), round(float(o["gamma"]), 4)))
print(mode, out)
