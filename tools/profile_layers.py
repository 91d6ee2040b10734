#!/usr/bin/env python3
"""Per-shape timing of the conv kernel families over one training step (HIP events around each launch).

    python tools/profile_layers.py [--batch 16] [--size 256] > gpurun_out/layers.txt
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import make_config  # noqa: E402
from behavior_driven_video_synthesis_amd import ops  # noqa: E402
from behavior_driven_video_synthesis_amd.experiments.shape_and_pose_net import ShapePoseNet, synthetic_batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--regressor", action="store_true")
args = ap.parse_args()
cfg = make_config(args)
tr = ShapePoseNet(cfg, device="cuda:0", total_steps=150000, vgg_synthetic=True)
batch = synthetic_batch(args.batch, args.size, "cuda:0", seed=42)
for _ in range(2):
    tr.train_fn(batch)
torch.cuda.synchronize()
ops.profile_start()
for _ in range(args.steps):
    tr.train_fn(batch)
torch.cuda.synchronize()
fam = ops.profile_stop(detail=True)
rows = sorted(fam.items(), key=lambda kv: -kv[1]["ms"])
tot = sum(v["ms"] for v in fam.values()) / args.steps
print(f"total conv ms/step {tot:.2f}")
print(f"{'kernel':18s} {'N':>3s} {'C1':>4s} {'C2':>4s} {'Hs':>4s} {'Ws':>4s} {'M':>4s} k s act {'n/step':>6s} {'ms/step':>8s} {'us/launch':>9s} {'TF/s':>6s} {'cum%':>5s}")
cum = 0.0
for key, v in rows:
    ms = v["ms"] / args.steps
    cum += ms
    name, n, c1, c2, hs, ws, m, k, s, act, kname = key
    print(f"{name:18s} {n:3d} {c1:4d} {c2:4d} {hs:4d} {ws:4d} {m:4d} {k} {s} {act:3d} {v['n'] // args.steps:6d} {ms:8.3f} "
          f"{1e3 * v['ms'] / v['n']:9.1f} {v['flop'] / (v['ms'] * 1e-3) / 1e12:6.1f} {100 * cum / tot:5.1f}  {kname}")
