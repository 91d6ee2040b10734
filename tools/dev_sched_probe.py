#!/usr/bin/env python3
"""Where the device-resident schedule costs time against the host-schedule eager step (one process, alternating blocks):
(1) eager host schedule on the default stream, (2) the same on a side stream with the waits the graph mode issues,
(3) device schedule (lr / step / dropout counter / gamma in device memory) eager, (4) the same without the dropout step
counter.    python tools/dev_sched_probe.py"""
import contextlib
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from behavior_driven_video_synthesis_amd import ops  # noqa: E402
from behavior_driven_video_synthesis_amd.experiments.shape_and_pose_net import ShapePoseNet, synthetic_batch  # noqa: E402

sys.argv = sys.argv[:1]
a = bench.parse()
cfg = bench.make_config(a)
with contextlib.redirect_stdout(sys.stderr):
    tr = ShapePoseNet(cfg, device="cuda:0", total_steps=150000, vgg_synthetic=True)
batch = synthetic_batch(16, 256, "cuda:0", seed=42)
side = torch.cuda.Stream()


def step_plain():
    tr.train_fn(batch)


def step_side():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        tr.train_fn(batch)
    cur.wait_stream(side)


def step_side_nowait():
    with torch.cuda.stream(side):
        tr.train_fn(batch)


def block(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


res = {}
for _ in range(6):
    tr.train_fn(batch)
res["1 eager host schedule, default stream"] = [round(block(step_plain), 2) for _ in range(3)]
res["2 eager host schedule, side stream + waits"] = [round(block(step_side), 2) for _ in range(3)]
res["2b eager host schedule, side stream, no waits"] = [round(block(step_side_nowait), 2) for _ in range(3)]
res["1 again"] = [round(block(step_plain), 2) for _ in range(3)]
tr.enable_hip_graph(capture=False)
res["3 eager device schedule (on its stream)"] = [round(block(step_plain), 2) for _ in range(3)]
orig = ops.set_dropout_step
ops.set_dropout_step = lambda c: orig(None)      # the trainer re-asserts its counter every step: hand the library None instead
res["4 device schedule, no dropout step counter"] = [round(block(step_plain), 2) for _ in range(3)]
ops.set_dropout_step = orig
print(json.dumps(res, indent=1))
