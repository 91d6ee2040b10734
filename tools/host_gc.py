#!/usr/bin/env python3
"""Host issue time of the bs-16 step with Python's cyclic GC enabled / disabled, alternating blocks (a full collection
walks ~180 k long-lived objects: 22 ms, tools/gc_objs.py).   python tools/host_gc.py"""
import contextlib
import gc
import os
import statistics
import sys
import time

os.environ["VUNET_GC_FREEZE"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import make_config, parse  # noqa: E402
from behavior_driven_video_synthesis_amd.experiments.shape_and_pose_net import ShapePoseNet, synthetic_batch  # noqa: E402

sys.argv = sys.argv[:1]
args = parse()
with contextlib.redirect_stdout(sys.stderr):
    tr = ShapePoseNet(make_config(args), device="cuda:0", total_steps=150000, vgg_synthetic=True)
batch = synthetic_batch(16, 256, "cuda:0", seed=42)
for _ in range(10):
    tr.train_fn(batch)
res = {True: [], False: []}
for rnd in range(4):
    for on in (True, False):
        gc.enable() if on else gc.disable()
        ts = []
        for _ in range(15):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tr.train_fn(batch)
            ts.append(1e3 * (time.perf_counter() - t0))
        res[on].append((round(statistics.median(ts), 2), round(statistics.mean(ts), 2), round(max(ts), 2)))
gc.enable()
for on in (True, False):
    print("gc", "enabled " if on else "disabled", "issue ms (median, mean, max) per block:", res[on])
