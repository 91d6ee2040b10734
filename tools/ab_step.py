#!/usr/bin/env python3
"""A/B of a runtime switch on the bs-16 training step inside ONE process (boxes differ by several per cent, so two bench
runs cannot resolve a 2 % change):  python tools/ab_step.py grad_passthrough   -> ms/step with the switch on / off,
alternating blocks of steps."""
import contextlib
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import make_config, parse  # noqa: E402
from behavior_driven_video_synthesis_amd import ops  # noqa: E402
from behavior_driven_video_synthesis_amd.experiments.shape_and_pose_net import ShapePoseNet, synthetic_batch  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "grad_passthrough"
sys.argv = sys.argv[:1]
args = parse()
with contextlib.redirect_stdout(sys.stderr):
    tr = ShapePoseNet(make_config(args), device="cuda:0", total_steps=150000, vgg_synthetic=True)
batch = synthetic_batch(16, 256, "cuda:0", seed=42)


def switch(on):
    if what == "grad_passthrough":
        ops.enable_grad_passthrough(on)
    elif what == "wn_batch":
        ops.enable_wn_batching(on)
    elif what == "wgrad_rowsplit":
        ops.set_tuning("wgrad_rowsplit", 0 if on else 1)
    elif what == "wgrad_rowsplit_all":
        ops.set_tuning("wgrad_rowsplit", 2 if on else 0)
    elif what == "l1_pool":
        ops.l1_pool_fusion["on"] = on
    elif what == "s2_fwd_h2":
        ops.set_tuning("s2_fwd_f32", 0 if on else 1)
    elif what == "target_overlap":
        tr._target_overlap = on
    elif what == "pack_overlap":
        tr._pack_overlap = on
    elif what == "relu_premask":
        ops.enable_relu_premask(on)
    elif what == "fused_parity":          # the stride-2 data gradient: one fused launch (on) / four per-parity launches
        ops.set_tuning("parity_launches", 0 if on else 1)
    else:
        raise SystemExit(f"unknown switch {what}")


res = {True: [], False: []}
for rnd in range(4):
    for on in (True, False):
        switch(on)
        for _ in range(3):
            tr.train_fn(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            tr.train_fn(batch)
        torch.cuda.synchronize()
        res[on].append(1e2 * (time.perf_counter() - t0))
switch(True)
for on in (True, False):
    print(what, "on " if on else "off", "ms/step per block:", [round(v, 2) for v in res[on]], "median", round(statistics.median(res[on]), 2))
