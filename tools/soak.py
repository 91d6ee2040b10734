import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from bench import make_config, parse
from behavior_driven_video_synthesis_amd.experiments.shape_and_pose_net import ShapePoseNet, synthetic_batch
sys.argv = [sys.argv[0]]
args = parse()
cfg = make_config(args)
cfg["training"]["n_init_batches"] = 4
tr = ShapePoseNet(cfg, device="cuda:0", total_steps=150000, vgg_synthetic=True)
if os.environ.get("SOAK_GRAPH", "1") == "1":
    tr.enable_hip_graph()      # the replayed step (bench.py's default); SOAK_GRAPH=0: eager
N = int(os.environ.get("SOAK_STEPS", "300"))
batches = [synthetic_batch(16, 256, "cuda:0", seed=s) for s in range(4)]
t0 = time.perf_counter()
for i in range(N):
    out = tr.train_fn(batches[i % 4])
    if i % max(N // 6, 1) == max(N // 6, 1) - 1:
        torch.cuda.synchronize()
        print(i + 1, f"loss {float(out['loss']):.3f} kl {float(out['kl_loss']):.3f} gamma {float(out['gamma']):.5f}",
              f"mem {torch.cuda.memory_allocated() / 2**30:.2f} GiB peak {torch.cuda.max_memory_allocated() / 2**30:.2f} reserved {torch.cuda.memory_reserved() / 2**30:.2f}",
              f"{(time.perf_counter() - t0) / (i + 1) * 1e3:.1f} ms/step", flush=True)
