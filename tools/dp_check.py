#!/usr/bin/env python3
"""Data-parallel correctness run of ShapePoseNet.train_fn on real GPUs (driven by tests/test_hip_dp_multi_gpu.py).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P \
        tools/dp_check.py --out DIR            # every rank trains on its shard of the global batch
    python tools/dp_check.py --single --world 2 --out DIR   # one rank, the concatenated global batch

Dropout is off and the posterior noise is injected per sample, so the N-rank run (mean of per-rank gradients, RCCL
all-reduce from backward's hooks, weight-gradient companion streams on) and the single-rank run on the concatenated
batch are the same computation up to summation order.  Each process writes its final parameters'
checksums and one full tensor to DIR/rank<r>.pt (single: DIR/single.pt).
"""
import argparse
import contextlib
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--out", required=True)
ap.add_argument("--single", action="store_true")
ap.add_argument("--world", type=int, default=0, help="--single: the world size whose global batch is concatenated")
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--batch", type=int, default=2, help="per-rank batch")
ap.add_argument("--size", type=int, default=64)
ap.add_argument("--graph", choices=["off", "capture", "eager", "capture-fail"], default="off",
                help="device-resident schedule: the step replayed from a captured hipGraph (all-reduces inside) / the same "
                     "schedule launched eagerly / a capture that is made to fail after the whole step has been recorded (the "
                     "trainer must fall back to eager issue with the averager's state intact).  The posterior noise is then "
                     "drawn inside the sampling kernels from the dropout seed base (ops.set_dropout_seed: the config seed, "
                     "offset per rank by the trainer), not from torch's generator")
a = ap.parse_args()

world = a.world if a.single else int(os.environ["WORLD_SIZE"])
rank = 0 if a.single else int(os.environ["RANK"])
local = 0 if a.single else int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
if not a.single:
    dist.init_process_group("nccl", device_id=dev)

from behavior_driven_video_synthesis_amd.experiments.shape_and_pose_net import DEFAULT_CONFIG, ShapePoseNet  # noqa: E402

cfg = copy.deepcopy(DEFAULT_CONFIG)
cfg["data"]["spatial_size"] = a.size
cfg["architecture"].update(nf_start=16, nf_max=32)
cfg["training"].update(dropout_prob=0.0, train_regressor=False, n_init_batches=1, lr=1e-3, gamma_step=1e-3,
                       information_max=5.0)
with contextlib.redirect_stdout(sys.stderr):
    tr = ShapePoseNet(cfg, device=dev, vgg_width_div=8, total_steps=100, vgg_synthetic=True)   # same seed on every rank
if a.graph != "off":
    tr.enable_hip_graph(capture=(a.graph != "eager"))
    if a.graph == "capture-fail":
        real_step = tr._step

        def step_that_fails_while_recording(*args, **kw):
            out_ = real_step(*args, **kw)   # the whole step is recorded first: hooks fired, host-side step counts advanced
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("injected capture failure (tools/dp_check.py --graph capture-fail)")
            return out_
        tr._step = step_that_fails_while_recording
nlat = cfg["architecture"]["n_latent_scales"]
lat = [(32, a.size // 2 ** (tr.vunet.n_scales - 1 - i)) for i in range(nlat)]   # (channels, width) of the latent scales


def global_batch(step):
    g = torch.Generator().manual_seed(1000 + step)
    n = world * a.batch
    x = torch.rand(n, 3, a.size, a.size, generator=g) * 2 - 1
    c = (torch.rand(n, 3, a.size, a.size, generator=g) < 0.05).float() * 2 - 1
    eps = [torch.randn(n, ch, w, w, generator=g) for ch, w in lat]
    return x, c, eps


losses = []
for step in range(a.steps):
    x, c, eps = global_batch(step)
    sl = slice(0, world * a.batch) if a.single else slice(rank * a.batch, (rank + 1) * a.batch)
    out = tr.train_fn({"pose_img": x[sl].to(dev), "stickman": c[sl].to(dev)},
                      None if a.graph != "off" else [e[sl].to(dev) for e in eps])
    losses.append(float(out["loss"]))
torch.cuda.synchronize()
sd = tr.vunet.state_dict()
res = {"losses": losses, "gamma": float(tr.gamma),
       "sums": {k: float(v.double().sum()) for k, v in sd.items()},
       "tensor": sd["dd.out_conv.conv.weight_v"].cpu(), "flat": [b.flat.cpu() for b in tr.optimizer.buckets],
       "adam_steps": [int(b.step) for b in tr.optimizer.buckets], "allreduce_ms": tr.averager.mean_allreduce_ms(), "backend": tr.averager.backend, "graphs": len(tr._graphs)}
os.makedirs(a.out, exist_ok=True)
torch.save(res, os.path.join(a.out, "single.pt" if a.single else f"rank{rank}.pt"))
if dist.is_initialized():
    from behavior_driven_video_synthesis_amd.parallel import shutdown_native_comm
    dist.barrier()
    shutdown_native_comm()
    dist.destroy_process_group()
