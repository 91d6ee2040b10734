#!/usr/bin/env python3
"""Which convolutions of a bs-16 training step still run their own |x|-maxima pass (vunet_absmax_partials) in the fp16 scheme:
caller, shapes of the sources, whether each carries a producer tag (needs an MI355X).  r02: 37 per step -- inputs produced by
the 3-channel / 1x1 / stride-2 kernels (no tags), two-source layers, gradients summed by the autograd engine."""
import sys, copy, contextlib, collections, traceback, torch
sys.path.insert(0, "/root/repo")
from behavior_driven_video_synthesis_amd import ops
from behavior_driven_video_synthesis_amd.experiments.shape_and_pose_net import DEFAULT_CONFIG, ShapePoseNet, synthetic_batch
cfg = copy.deepcopy(DEFAULT_CONFIG)
with contextlib.redirect_stdout(sys.stderr):
    tr = ShapePoseNet(cfg, device="cuda:0", total_steps=150000, vgg_synthetic=True)
batch = synthetic_batch(16, 256, "cuda:0", seed=42)
for _ in range(3): tr.train_fn(batch)
orig = ops.absmax_partials
census = collections.Counter()
def rec(x1, x2=None):
    st = traceback.extract_stack(limit=6)
    where = "/".join(f.name for f in st[:-1] if f.name in ("forward", "backward", "get_dy_amax", "_conv_gather", "dgrad", "weight_gradients"))
    t1 = ops._tagged_amax(x1) is not None
    t2 = None if x2 is None else (ops._tagged_amax(x2) is not None)
    census[(where, tuple(x1.shape), None if x2 is None else tuple(x2.shape), t1, t2)] += 1
    return orig(x1, x2)
ops.absmax_partials = rec
tr.train_fn(batch)
torch.cuda.synchronize()
for k, v in sorted(census.items(), key=lambda kv: -kv[0][1][1] * kv[0][1][2] * kv[0][1][3]):
    print(v, k)
print("total", sum(census.values()))
