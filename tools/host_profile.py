import cProfile, pstats, sys, io, torch
sys.path.insert(0, "/root/repo")
from behavior_driven_video_synthesis_amd.experiments.shape_and_pose_net import DEFAULT_CONFIG, ShapePoseNet, synthetic_batch
import copy, contextlib
cfg = copy.deepcopy(DEFAULT_CONFIG)
with contextlib.redirect_stdout(sys.stderr):
    tr = ShapePoseNet(cfg, device="cuda:0", total_steps=150000, vgg_synthetic=True)
batch = synthetic_batch(16, 256, "cuda:0", seed=42)
for _ in range(5): tr.train_fn(batch)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10): tr.train_fn(batch)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
