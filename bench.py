#!/usr/bin/env python3
"""bench.py -- frames/sec of one VUnet shape-and-posture training step on MI355X.

A "step" is one pass of the hot path over one synthetic batch: VunetAlter 256x256 forward, VGG19
perceptual loss (target pass + prediction pass) + KL, backward (dgrad + wgrad), fused Adam -- per-GPU
batch 16 (BASELINE.json configs[1]; weak scaling: global batch = 16 * N).  Dropout 0.05 is on, the
regressor side loop is off (flag --regressor turns it on; it adds 5 encoder forwards per step and no
gradient to the VUnet), VGG19 weights are seeded-synthetic (no network for the pretrained ones).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events around every conv-family
launch in an extra instrumented region after the timed one (so the events do not perturb `value`);
`cpu_baseline` times the CPU oracle (oracle/vunet_oracle.py, a port) on a bounded sample of the same
workload on the host cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
FLOP_PER_FRAME = 275.6e9       # VUnet f+b 130.0 GF + perceptual (target fwd, pred fwd+dgrad) 145.6 GF


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--regressor", action="store_true", help="run the regressor side loop as the reference config does")
    ap.add_argument("--gan", action="store_true",
                    help="add the PartDiscriminator adversarial term + one discriminator step (not in the reference loop)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=1)
    ap.add_argument("--cpu-steps", type=int, default=2)
    return ap.parse_args()


def make_config(args):
    from behavior_driven_video_synthesis_amd.experiments.shape_and_pose_net import DEFAULT_CONFIG
    import copy
    cfg = copy.deepcopy(DEFAULT_CONFIG)
    cfg["data"]["spatial_size"] = args.size
    cfg["training"]["batch_size"] = args.batch
    cfg["training"]["train_regressor"] = bool(args.regressor)
    cfg["training"]["gan"]["enabled"] = bool(getattr(args, "gan", False))
    return cfg


def cpu_baseline(args, cfg):
    """The CPU oracle (a port of the reference path) timed on this host's cores on a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from oracle import vunet_oracle as O
    from behavior_driven_video_synthesis_amd.models.vunets import VunetAlter
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))  # PyTorch-CPU conv stops scaling (and thrashes) far below 256 threads
    torch.set_num_threads(cores)
    kw = dict(cfg["architecture"])
    kw.update(cfg["data"])
    kw["dropout_prob"] = 0.0
    torch.manual_seed(42)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):
        net = VunetAlter(**kw)  # host-side parameter container only: gives the reference's default init + key layout
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    vsd = O.make_synthetic_vgg19(seed=1234)
    opt = torch.optim.Adam([{"params": [v for k, v in sd.items() if k.startswith(n + ".")], "name": n}
                            for n in ("eu", "ed", "du", "dd")], lr=5e-4, betas=(0.5, 0.9))
    b, s = args.cpu_batch, args.size
    g = torch.Generator().manual_seed(42)
    x = torch.rand(b, 3, s, s, generator=g) * 2 - 1
    c = (torch.rand(b, 3, s, s, generator=g) < 0.05).float() * 2 - 1
    times, budget, t_start = [], 30.0, time.perf_counter()
    for it in range(1 + args.cpu_steps):
        t0 = time.perf_counter()
        loss, ll, kl, _ = O.train_step_losses(sd, kw, vsd, [1.0] * 6, x, c, x, None, 0.0, it + 10, 4)
        opt.zero_grad()
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start + times[-1] > budget:  # bounded sample: stop before exceeding ~30 s
            break
    t = min(times[1:]) if len(times) > 1 else times[0]
    return {"value": b / t, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"fastest of {max(len(times) - 1, 1)} timed step(s) ({len(times)} run, ~30 s budget) of the same "
                      f"training step at {s}x{s}, batch {b}, dropout off, PyTorch-CPU fp32 oracle, {cores} threads"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("VUNET_DP_FORCE") == "1":
        dist.init_process_group("nccl", device_id=device)
    from behavior_driven_video_synthesis_amd import ops
    from behavior_driven_video_synthesis_amd.experiments.shape_and_pose_net import ShapePoseNet, synthetic_batch

    cfg = make_config(args)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):  # the constructors print like the reference does; stdout is for the JSON line
        trainer = ShapePoseNet(cfg, device=device, total_steps=150000, vgg_synthetic=True)
    batch = synthetic_batch(args.batch, args.size, device, seed=42, with_regressor=args.regressor, rank=rank)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.train_fn(batch)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = trainer.train_fn(batch)
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_val = float(out["loss"])
    assert loss_val == loss_val, "loss is NaN"

    result = {
        "metric": "frames/sec VUnet 256x256 bs=16 fwd+bwd", "value": world * args.batch * args.steps / elapsed,
        "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"Human3.6m shape_and_pose_net VunetAlter {args.size}x{args.size} per-GPU bs={args.batch} "
                               "fwd+bwd, VGG19 perceptual + KL loss, fused Adam, dropout 0.05"
                               + (", regressor side loop on" if args.regressor else ", regressor side loop off")
                               + (", adversarial term on" if args.gan else "")
                               + ", seeded-synthetic VGG19 weights",
                   "global_batch": world * args.batch, "parallelism": f"dp{world}",
                   "flop_per_frame": FLOP_PER_FRAME, "final_loss": loss_val,
                   "hip_streams": 1 if trainer.vunet._side_stream is None else 4},
    }

    if not args.no_roofline:
        # instrumented region: HIP events around every conv-family launch (same stream as the kernels).
        # Every rank runs these steps (train_fn contains the gradient all-reduce); only rank 0 reports.
        # The timed region runs the pose encoder on a second HIP stream beside the appearance encoder; here both are put back on one
        # stream so that every event pair times its kernel alone on the GPU (exclusive per-kernel durations).
        prof_steps = max(1, min(3, args.steps))
        side_stream = trainer.vunet._side_stream
        trainer.vunet._side_stream = None
        wgrad_streams = ops._wgrad_streams["on"]
        ops.enable_wgrad_streams(False)
        sync_all()
        ops.profile_start()
        for _ in range(prof_steps):
            trainer.train_fn(batch)
        sync_all()
        trainer.vunet._side_stream = side_stream
        ops.enable_wgrad_streams(wgrad_streams)
    if rank == 0 and not args.no_roofline:
        recs = ops._prof["recs"][:]                       # raw (key, flop, ev0, ev1) records of the instrumented steps
        fam = ops.profile_stop()                           # per family: conv_gather_fwd / conv_gather_dgrad / conv_wgrad
        ops._prof["recs"] = recs
        kern = ops.profile_stop(by_kernel=True)            # per kernel instantiation, rocprofv3 spelling
        tot_ms = sum(v["ms"] for v in fam.values())
        dom = max(kern, key=lambda k: kern[k]["ms"])
        ach = kern[dom]["flop"] / (kern[dom]["ms"] * 1e-3) / 1e12
        result["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS,
                              "unit": "TFLOP/s", "frac": ach / FP32_MFMA_PEAK_TFLOPS, "traffic": None,
                              "launches_per_step": kern[dom]["n"] // prof_steps,
                              "avg_launch_us": 1e3 * kern[dom]["ms"] / kern[dom]["n"],
                              "algorithmic_gflop_per_launch": kern[dom]["flop"] / kern[dom]["n"] / 1e9,
                              "share_of_conv_time": kern[dom]["ms"] / tot_ms,
                              "families": {k: {"ms_per_step": v["ms"] / prof_steps, "launches_per_step": v["n"] // prof_steps,
                                               "tflops": v["flop"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0.0}
                                           for k, v in fam.items()},
                              "kernels": {k: {"ms_per_step": round(v["ms"] / prof_steps, 3),
                                              "avg_launch_us": round(1e3 * v["ms"] / v["n"], 1),
                                              "tflops": round(v["flop"] / (v["ms"] * 1e-3) / 1e12, 1)}
                                          for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["ms"])[:20]},
                              "conv_ms_per_step": tot_ms / prof_steps,
                              "whole_step_tflops": FLOP_PER_FRAME * args.batch / (1e-3 * result["ms_per_step"]) / 1e12}
        # HBM traffic per launch of the dominant kernel: PMC passes collected separately (tools/pmc_summary.py)
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            ent = pmc["kernels"].get(dom)
            if ent is not None:
                result["roofline"]["traffic"] = ent["hbm_bytes_per_launch"]
                result["roofline"]["traffic_source"] = "profiles/r01_pmc_traffic.json: " + pmc["correction"]
        except (OSError, ValueError, KeyError):
            pass
    elif not args.no_roofline:
        ops.profile_stop()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args, cfg)
    if rank == 0:
        print(json.dumps(result))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
