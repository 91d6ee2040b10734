#!/usr/bin/env python3
"""Eager issue vs hipGraph replay of the bs-16 training step inside ONE process (boxes differ by several per cent).

    python tools/graph_probe.py [--regressor] [--gan] [--batch 16] [--size 256] [--blocks 3]

Prints one JSON line: ms/step and host issue time of (i) the eager host-schedule step, (ii) the eager device-schedule
step on the capture stream, (iii) the replayed graph; the number of graph nodes by type when the runtime reports them."""
import argparse
import contextlib
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from behavior_driven_video_synthesis_amd.experiments.shape_and_pose_net import ShapePoseNet, synthetic_batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--regressor", action="store_true")
ap.add_argument("--gan", action="store_true")
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--blocks", type=int, default=3)
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
cfg = bench.make_config(a)
with contextlib.redirect_stdout(sys.stderr):
    tr = ShapePoseNet(cfg, device="cuda:0", total_steps=150000, vgg_synthetic=True)
batch = synthetic_batch(a.batch, a.size, "cuda:0", seed=42, with_regressor=a.regressor)


def block(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tr.train_fn(batch)
    issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n, 1e3 * issue / n


def one_issue():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train_fn(batch)
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return 1e3 * dt


def measure(tag, res):
    for _ in range(3):
        tr.train_fn(batch)
    ms = [block(a.steps) for _ in range(a.blocks)]
    res[tag] = {"ms_per_step": round(statistics.median(m[0] for m in ms), 3), "blocks": [round(m[0], 2) for m in ms],
                "host_issue_ms_empty_queue": round(statistics.median(one_issue() for _ in range(3)), 3)}


res = {"batch": a.batch, "size": a.size, "regressor": a.regressor, "gan": a.gan,
       "capture_forks": os.environ.get("VUNET_CAPTURE_FORKS", "1"), "wn_batch_capture": os.environ.get("VUNET_WN_BATCH_CAPTURE", "1")}
for _ in range(6):
    tr.train_fn(batch)      # past the init batches
measure("eager_host_schedule", res)
tr.enable_hip_graph(capture=False)
measure("eager_device_schedule", res)
tr.enable_hip_graph(capture=True)
measure("graph_replay", res)
res["captured"] = bool(tr._graphs)
res["final_loss"] = float(tr.train_fn(batch)["loss"])
print(json.dumps(res))
