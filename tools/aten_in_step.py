#!/usr/bin/env python3
"""Which ATen operators still launch device work inside one training step (torch.profiler, 3 steps)?

    python tools/aten_in_step.py            (needs an MI355X)
"""
import contextlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from bench import make_config, parse  # noqa: E402
from behavior_driven_video_synthesis_amd.experiments.shape_and_pose_net import ShapePoseNet, synthetic_batch  # noqa: E402

sys.argv = sys.argv[:1]
args = parse()
with contextlib.redirect_stdout(sys.stderr):
    tr = ShapePoseNet(make_config(args), device="cuda:0", total_steps=150000, vgg_synthetic=True)
tr.enable_hip_graph(capture=False)   # the device-resident schedule of the captured step, issued eagerly so the profiler sees it
batch = synthetic_batch(16, 256, "cuda:0", seed=42)
for _ in range(3):
    tr.train_fn(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(3):
        tr.train_fn(batch)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_stack_n=6) if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.count)
for e in rows[:40]:
    print(f"{e.count / 3:7.1f}/step  dev {e.device_time_total / 3:9.1f} us/step  {e.key}")
    for fr in e.stack[:5]:
        if "behavior_driven" in fr or "bench" in fr:
            print("            ", fr)

# ---- who calls them: top-level ATen calls with device time, by the nearest frame of this package
import collections  # noqa: E402
cnt = collections.Counter()
for e in prof.events():
    if not e.name.startswith("aten::") or e.device_time_total <= 0:
        continue
    if e.cpu_parent is not None and e.cpu_parent.name.startswith("aten::"):
        continue
    frames = [f for f in (e.stack or []) if "behavior_driven" in f or "bench" in f]
    cnt[(e.name, frames[0].strip()[-110:] if frames else "?")] += 1
print()
for (name, where), n in cnt.most_common(40):
    print(f"{n / 3:6.1f}/step  {name:18s} {where}")
