import sys, os, contextlib
sys.path.insert(0, "/root/repo")
import torch
from bench import make_config, parse
from behavior_driven_video_synthesis_amd.experiments.shape_and_pose_net import ShapePoseNet, synthetic_batch
sys.argv = sys.argv[:1]
args = parse()
with contextlib.redirect_stdout(sys.stderr):
    tr = ShapePoseNet(make_config(args), device="cuda:0", total_steps=150000, vgg_synthetic=True)
batch = synthetic_batch(16, 256, "cuda:0", seed=42)
graph = len(sys.argv) > 1 or os.environ.get("G") == "1"
if os.environ.get("G") == "1":
    tr.enable_hip_graph()
for _ in range(12):
    tr.train_fn(batch)
torch.cuda.synchronize()
print("graph" if os.environ.get("G") == "1" else "eager", "allocated GB", torch.cuda.max_memory_allocated() / 2**30, "reserved GB", torch.cuda.max_memory_reserved() / 2**30)
