#!/usr/bin/env python3
"""bench.py -- frames/sec of one VUnet shape-and-posture training step on MI355X.

A "step" is one pass of the hot path over one synthetic batch: VunetAlter 256x256 forward, VGG19
perceptual loss (target pass + prediction pass) + KL, backward (dgrad + wgrad), fused Adam -- per-GPU
batch 16 (BASELINE.json configs[1]; weak scaling: global batch = 16 * N).  Dropout 0.05 is on, the
regressor side loop is off (flag --regressor turns it on; it adds 5 encoder forwards per step and no
gradient to the VUnet), VGG19 weights are seeded-synthetic (no network for the pretrained ones), the
stickman input is drawn by the GPU rasteriser from synthetic 17-joint skeletons.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N --steps K --warmup W          # N > 1: bench.py starts the N ranks itself (below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W              # ... or is started as one rank of N

Rank 0 prints ONE JSON line.
  --gpus N      replaces nn.DataParallel (experiments/shape_and_pose_net.py:213-214,223-224,230-233): one process per GPU,
                gradient buckets all-reduced over RCCL while backward runs.  Called WITHOUT a torchrun environment and
                with N > 1, bench.py starts `python -m torch.distributed.run --nproc-per-node N ... bench.py <same args>`
                as a fresh CHILD process before anything touches the GPU (never an exec of this process), relays rank
                0's JSON line and the child's exit code, and fails non-zero if fewer than N GPUs are visible or the line's
                `n_gpus` / `rccl_world_size` is not N.  `VUNET_DP_FORCE=1 python bench.py --gpus 1` takes the same route
                with one rank (launcher + one-rank RCCL communicator on a 1-GPU box).
  roofline      measured live with HIP events around every conv-family launch in an extra instrumented region
                after the timed one (so the events do not perturb `value`); `traffic` comes from the committed
                rocprofv3 PMC summary profiles/<round>_pmc_traffic.json (tools/profile.sh), named in
                `traffic_source` with the commit it was collected at, and is null if the dominant kernel is not in it.
  cpu_baseline  the CPU oracle (oracle/vunet_oracle.py, a port of the reference path, pinned to the reference by
                tests/golden) timed on this host's cores as BASELINE.md section 4 defines it: the SAME bs-16 256^2
                step, 1 warm-up + 3 timed steps, median -- on min(host cores, --cpu-threads = 64) threads (with hundreds
                of threads PyTorch-CPU thrashes on this path: a run with all 256 did not finish in 35 minutes), inside
                a stated time budget and in a child process with a hard wall-clock limit (if the budget bites the bs-1
                figure is reported as well and said so); plus the config-1 plumbing row (Market 128^2, bs 2, 30-channel
                64x64 appearance input).  Rank 0, N = 1 only.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same guide: dense bf16 MFMA
X6_PRODUCTS = 6                 # bf16 MFMAs per fp32-accurate MAC block in the split kernels (csrc/conv_x6_kernel.h)
FLOP_PER_FRAME = 275.6e9        # VUnet f+b 130.0 GF + perceptual (target fwd, pred fwd+dgrad) 145.6 GF
PMC_TRAFFIC_RENDER = ["profiles/r06_pmc_traffic_render.json", "profiles/r05_pmc_traffic_render.json", "profiles/r04_pmc_traffic_render.json", "profiles/r03_pmc_traffic_render.json"]
PMC_TRAFFIC_SEQ = ["profiles/r06_pmc_traffic_seq.json", "profiles/r05_pmc_traffic_seq.json"]
PMC_TRAFFIC_SEQ_TRAIN = ["profiles/r06_pmc_traffic_seq_train.json"]
PMC_TRAFFIC = ["profiles/r06_pmc_traffic.json", "profiles/r05_pmc_traffic.json", "profiles/r04_pmc_traffic.json", "profiles/r03_pmc_traffic.json", "profiles/r02_pmc_traffic.json",
               "profiles/r01_pmc_traffic.json"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--regressor", action="store_true", help="run the regressor side loop as the reference config does")
    ap.add_argument("--gan", action="store_true",
                    help="add the PartDiscriminator adversarial term + one discriminator step (not in the reference loop)")
    ap.add_argument("--precision", choices=["h2", "x6", "f32"], default="h2",
                    help="h2 / x6: fp32-accurate split convolution kernels where they apply (two scaled fp16 terms, three "
                         "products / three bf16 terms, six products); f32: fp32-input MFMA everywhere")
    ap.add_argument("--hip-graph", choices=["auto", "on", "off"], default="auto",
                    help="replay the step from a captured multi-stream hipGraph (the host then issues one launch per step "
                         "instead of ~600).  auto: on wherever the trainer can capture the configuration, reported in "
                         "config.hip_graph; the config-1 row always reports both")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-config1", action="store_true", help="skip the config-1 (Market 128^2, bs 2) plumbing rows")
    ap.add_argument("--no-render", action="store_true", help="skip the config-5 render-loop row")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the `variants` rows (the step with the regressor side loop / the adversarial term / both, and the "
                         "fp32-input-MFMA precision, 10 replayed steps each on this box)")
    ap.add_argument("--cpu-budget", type=float, default=100.0, help="seconds of CPU work allowed for the bs-16 baseline")
    ap.add_argument("--cpu-threads", type=int, default=64, help="cap on the CPU baseline's threads")
    ap.add_argument("--cpu-child", default=None, help=argparse.SUPPRESS)
    return ap.parse_args()


def make_config(args):
    from behavior_driven_video_synthesis_amd.experiments.shape_and_pose_net import DEFAULT_CONFIG
    import copy
    cfg = copy.deepcopy(DEFAULT_CONFIG)
    cfg["data"]["spatial_size"] = args.size
    cfg["training"]["batch_size"] = args.batch
    cfg["training"]["train_regressor"] = bool(getattr(args, "regressor", False))
    cfg["training"]["gan"]["enabled"] = bool(getattr(args, "gan", False))
    return cfg


def market_config():
    """BASELINE config 1: Market1501, 128x128, bs 2, 30-channel 64x64 appearance input (README.md:103-110)."""
    from behavior_driven_video_synthesis_amd.experiments.shape_and_pose_net import DEFAULT_CONFIG
    import copy
    cfg = copy.deepcopy(DEFAULT_CONFIG)
    cfg["data"].update(dataset="Market", spatial_size=128, box_factor=1, bottleneck_factor=1, inplane_normalize=True)
    cfg["training"].update(batch_size=2, train_regressor=False)
    return cfg


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def _cpu_steps(cfg, batch, n_channels_x, warmup, timed, budget_s):
    """Time the CPU oracle's training step (VUnet + VGG19 perceptual + KL + torch.optim.Adam) -> list of step times."""
    import contextlib
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from oracle import vunet_oracle as O
    from behavior_driven_video_synthesis_amd.models.vunets import VunetAlter
    kw = dict(cfg["architecture"])
    kw.update(cfg["data"])
    kw["dropout_prob"] = 0.0
    torch.manual_seed(42)
    with contextlib.redirect_stdout(sys.stderr):
        net = VunetAlter(n_channels_x=n_channels_x, **kw)  # host-side parameter container: the reference's default init + keys
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    vsd = O.make_synthetic_vgg19(seed=1234)
    opt = torch.optim.Adam([{"params": [v for k, v in sd.items() if k.startswith(n + ".")], "name": n}
                            for n in ("eu", "ed", "du", "dd")], lr=5e-4, betas=(0.5, 0.9))
    target = batch["pose_img"].cpu()
    c = batch["stickman"].cpu()
    x = batch.get("pose_img_inplane", batch["pose_img"]).cpu()
    times, t_start = [], time.perf_counter()
    for it in range(warmup + timed):
        t0 = time.perf_counter()
        img, means, logstds, _ = O.vunet_alter_forward(sd, kw, x, c, None, n_channels_x=n_channels_x)
        ld = O.vgg_loss(vsd, [1.0] * 6, target, img)
        loss = torch.stack(list(ld.values()), dim=0).sum() + 1e-3 * O.compute_kl_with_prior(means, logstds)
        opt.zero_grad()
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
        spent = time.perf_counter() - t_start
        if it + 1 >= warmup + 1 and spent + times[-1] > budget_s:   # the next step would overrun the budget
            break
    return times[warmup:] if len(times) > warmup else times


def cpu_baseline_child(path):
    """Child-process body (no GPU is touched here): time the oracle on the batches the parent saved to ``path``."""
    job = torch.load(path)
    cores = job["threads"]
    torch.set_num_threads(cores)
    cfg, batch, budget = job["cfg"], job["batch"], job["budget"]
    b, s = batch["pose_img"].shape[0], batch["pose_img"].shape[-1]
    one = {k: v[:1] for k, v in batch.items()}
    t1 = _cpu_steps(cfg, one, 3, 1, 2, 30.0)                      # cheap probe: sizes the bs-16 run against the budget
    est = b * statistics.median(t1) * 0.8
    n_timed = max(0, min(3, int(budget / max(est, 1e-3)) - 1))
    out = {"unit": "frames/s", "cores": cores, "kind": "port"}
    b1 = {"value": 1.0 / statistics.median(t1), "unit": "frames/s", "sample": f"batch 1, median of {len(t1)} timed step(s) after 1 warm-up"}
    if n_timed >= 1:
        t16 = _cpu_steps(cfg, batch, 3, 1, n_timed, budget)
        out.update(value=b / statistics.median(t16), step_seconds=[round(t, 3) for t in t16],
                   sample=f"median of {len(t16)} timed step(s) after 1 warm-up of the SAME training step (VunetAlter {s}x{s} "
                          f"batch {b} fwd+bwd, VGG19 perceptual + KL, torch.optim.Adam; dropout off) on the PyTorch-CPU fp32 "
                          f"oracle, {cores} threads, budget {budget:.0f} s")
        if len(t16) < 3:
            out["batch1"] = dict(b1, note=f"reported as well because only {len(t16)} bs-{b} step(s) fit the budget")
    else:
        out.update(value=b1["value"], sample=b1["sample"] + f" -- a bs-{b} step is estimated at {est:.0f} s, over the "
                   f"{budget:.0f} s budget; same step definition, {cores} threads")
    if job.get("cfg1") is not None:
        t = _cpu_steps(job["cfg1"], job["batch1"], 30, 2, 10, 40.0)
        out["config1"] = {"value": job["batch1"]["pose_img"].shape[0] / statistics.median(t), "unit": "frames/s",
                          "sample": f"BASELINE config 1 (Market 128x128, batch 2, x = 30x64x64): median of {len(t)} timed "
                                    "step(s) after 2 warm-ups, same oracle / threads"}
    try:
        out["behavior"] = _cpu_behavior(cores)
    except Exception as e:   # noqa: BLE001 -- the headline baseline must not depend on this extra row
        out["behavior"] = {"error": f"{type(e).__name__}: {e}"}
    try:
        out["behavior_train"] = _cpu_behavior_train(cores)
    except Exception as e:   # noqa: BLE001
        out["behavior_train"] = {"error": f"{type(e).__name__}: {e}"}
    print("CPU_BASELINE " + json.dumps(out))


def _cpu_behavior(cores, rows=16, blocks=3, frames=50):
    """The behaviour front half of config 5 on the oracle (PyTorch-CPU fp32), bounded: ``blocks`` of the flow's 15 blocks at the
    reference width (the pass is 15 identical blocks in sequence: time x 5), and the full 50-step decoder roll-out."""
    from oracle import behavior_oracle as B
    g = torch.Generator().manual_seed(3)
    c, mid, c1 = 1024, 2048, 512
    sd = {}
    for i in range(blocks):
        q = f"flow.sub_layers.{i}"
        sd[f"{q}.norm_layer.loc"] = 0.1 * torch.randn(1, c, 1, 1, generator=g)
        sd[f"{q}.norm_layer.scale"] = 0.7 + 0.3 * torch.rand(1, c, 1, 1, generator=g)
        perm = torch.randperm(c, generator=g)
        sd[f"{q}.shuffle.forward_shuffle_idx"], sd[f"{q}.shuffle.backward_shuffle_idx"] = perm, torch.argsort(perm)
        for kind in ("s", "t"):
            for j in range(2):
                dims = [(mid, c1), (mid, mid), (mid, mid), (c1, mid)]
                for li, (o, k) in enumerate(dims):
                    gain = 0.1 if (kind == "s" and li == 3) else 1.0
                    sd[f"{q}.coupling.{kind}.{j}.main.{2 * li}.weight"] = gain * torch.randn(o, k, generator=g) * (1.5 / k) ** 0.5
                    sd[f"{q}.coupling.{kind}.{j}.main.{2 * li}.bias"] = 0.1 * torch.randn(o, generator=g)
    z = torch.randn(rows, c, generator=g)
    hid, n = 1024, 51
    dec = {"decoder.rnn.weight_ih": torch.randn(4 * hid, n, generator=g) / hid ** 0.5,
           "decoder.rnn.weight_hh": torch.randn(4 * hid, hid, generator=g) / hid ** 0.5,
           "decoder.rnn.bias_ih": 0.1 * torch.randn(4 * hid, generator=g), "decoder.rnn.bias_hh": 0.1 * torch.randn(4 * hid, generator=g),
           "decoder.n_out.weight": 0.05 * torch.randn(n, hid, generator=g) / hid ** 0.5, "decoder.n_out.bias": 0.1 * torch.randn(n, generator=g)}
    b, x = torch.randn(rows, hid, generator=g), 0.5 * torch.randn(rows, frames, n, generator=g)

    def med(fn, reps=5):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts)
    with torch.no_grad():
        t_flow = med(lambda: B.flow_reverse(sd, z)) * (15.0 / blocks)
        t_dec = med(lambda: B.generate_seq(dec, b, x, frames, frames - 1))
    return {"flow_reverse_ms": 1e3 * t_flow, "decode_ms": 1e3 * t_dec, "rows": rows, "cores": cores, "kind": "port",
            "sample": f"oracle/behavior_oracle.py: flow reverse of {blocks} of the 15 blocks at 1024 / 2048 (x {15 // blocks}: the pass is "
                      f"15 such blocks in sequence) and the full {frames}-step decoder roll-out at 1024 hidden, {rows} rows, median of 5"}


def _cpu_behavior_train(cores, rows=64, blocks=2, frames=50):
    """BASELINE configs[3] on the oracle (PyTorch-CPU fp32: torch.autograd + torch.optim.Adam), bounded: the flow stage's step on
    ``blocks`` of the 15 blocks at the reference width (x 15 / blocks: the blocks are identical and sequential), and one full
    cVAE step (1024 hidden, 51 dims, 50 frames), batch 64."""
    from oracle import behavior_oracle as B
    g = torch.Generator().manual_seed(4)
    c, mid, c1 = 1024, 2048, 512
    sd = {}
    for i in range(blocks):
        q = f"flow.sub_layers.{i}"
        sd[f"{q}.norm_layer.loc"] = 0.1 * torch.randn(1, c, 1, 1, generator=g)
        sd[f"{q}.norm_layer.scale"] = 0.7 + 0.3 * torch.rand(1, c, 1, 1, generator=g)
        perm = torch.randperm(c, generator=g)
        sd[f"{q}.shuffle.forward_shuffle_idx"], sd[f"{q}.shuffle.backward_shuffle_idx"] = perm, torch.argsort(perm)
        for kind in ("s", "t"):
            for j in range(2):
                for li, (o, k) in enumerate([(mid, c1), (mid, mid), (mid, mid), (c1, mid)]):
                    gain = 0.1 if (kind == "s" and li == 3) else 1.0
                    sd[f"{q}.coupling.{kind}.{j}.main.{2 * li}.weight"] = gain * torch.randn(o, k, generator=g) * (1.5 / k) ** 0.5
                    sd[f"{q}.coupling.{kind}.{j}.main.{2 * li}.bias"] = 0.1 * torch.randn(o, generator=g)
    opt = B.flow_optimizer(sd, 4.5e-7 * rows, 0.0)
    bs = torch.randn(rows, c, generator=g)
    B.flow_train_step(sd, opt, bs)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        B.flow_train_step(sd, opt, bs)
        ts.append(time.perf_counter() - t0)
    t_flow = statistics.median(ts) * (15.0 / blocks)
    del sd, opt
    hid, n = 1024, 51
    net = {"decoder.rnn.weight_ih": torch.randn(4 * hid, n, generator=g) / hid ** 0.5,
           "decoder.rnn.weight_hh": torch.randn(4 * hid, hid, generator=g) / hid ** 0.5,
           "decoder.rnn.bias_ih": 0.1 * torch.randn(4 * hid, generator=g), "decoder.rnn.bias_hh": 0.1 * torch.randn(4 * hid, generator=g),
           "decoder.n_out.weight": 0.05 * torch.randn(n, hid, generator=g) / hid ** 0.5, "decoder.n_out.bias": 0.1 * torch.randn(n, generator=g),
           "b_enc.rnn.weight_ih_l0": torch.randn(4 * hid, n, generator=g) / hid ** 0.5,
           "b_enc.rnn.weight_hh_l0": torch.randn(4 * hid, hid, generator=g) / hid ** 0.5,
           "b_enc.rnn.bias_ih_l0": 0.1 * torch.randn(4 * hid, generator=g), "b_enc.rnn.bias_hh_l0": 0.1 * torch.randn(4 * hid, generator=g)}
    for h in ("mu_fn", "std_fn"):
        net[f"b_enc.{h}.beta"], net[f"b_enc.{h}.gamma"] = torch.zeros(1, hid, 1, 1), torch.ones(1, hid, 1, 1)
        net[f"b_enc.{h}.conv.bias"], net[f"b_enc.{h}.conv.weight_g"] = torch.zeros(hid), torch.ones(hid, 1, 1, 1)
        net[f"b_enc.{h}.conv.weight_v"] = torch.randn(hid, hid, 1, 1, generator=g) / hid ** 0.5
    vopt = B.behavior_optimizer(net, 1e-4)
    kps, eps = 0.5 * torch.randn(rows, frames + 1, n, generator=g), torch.randn(rows, hid, generator=g)
    B.cvae_train_step(net, vopt, kps, eps, 0.0, 2.5, 1e-5, 100.0)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        B.cvae_train_step(net, vopt, kps, eps, 0.0, 2.5, 1e-5, 100.0)
        ts.append(time.perf_counter() - t0)
    t_vae = statistics.median(ts)
    return {"flow_stage": {"value": rows / t_flow, "unit": "samples/s", "step_s": t_flow},
            "cvae_stage": {"value": rows / t_vae, "unit": "sequences/s", "step_s": t_vae}, "rows": rows, "cores": cores, "kind": "port",
            "sample": f"oracle/behavior_oracle.py (torch.autograd + torch.optim.Adam): the flow stage's step on {blocks} of the 15 blocks at "
                      f"1024 / 2048 (x {15.0 / blocks:g}: the pass is 15 such blocks in sequence) and one full cVAE step (1024 hidden, 51 "
                      f"dims, {frames} frames), batch {rows}, median of 3 after 1 warm-up"}


def cpu_baseline(args, cfg, batch, cfg1, batch1):
    """Runs in a fresh child process with a hard wall-clock limit: a CPU step that thrashes (it does with hundreds of
    threads on this path) can then cost the budget, not the run."""
    import subprocess
    import tempfile
    cores = host_cores()
    threads = max(1, min(cores, args.cpu_threads))   # PyTorch-CPU conv stops scaling -- and on the small maps thrashes -- far below 256 threads
    cpu = lambda d: None if d is None else {k: v.detach().cpu() for k, v in d.items()}
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "job.pt")
        torch.save({"cfg": cfg, "batch": cpu(batch), "cfg1": cfg1, "batch1": cpu(batch1), "budget": args.cpu_budget,
                    "threads": threads}, path)
        limit = args.cpu_budget + 120.0
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-child", path], stdout=subprocess.PIPE,
                             stderr=subprocess.DEVNULL, text=True)
        try:
            so, _ = p.communicate(timeout=limit)
        except subprocess.TimeoutExpired:
            p.kill()
            p.communicate()
            return {"value": None, "unit": "frames/s", "cores": threads, "kind": "port",
                    "sample": f"CPU baseline child exceeded its hard limit of {limit:.0f} s and was stopped"}
    for line in so.splitlines():
        if line.startswith("CPU_BASELINE "):
            out = json.loads(line[len("CPU_BASELINE "):])
            out["host_cores"] = cores
            return out
    return {"value": None, "unit": "frames/s", "cores": threads, "kind": "port", "sample": "CPU baseline child failed"}


def roofline_entry(kern, fam, dom, tot_ms, prof_steps, ms_per_step, batch):
    is_x6 = "x6" in dom
    is_h2 = "h2" in dom or "conv_p2" in dom   # (p2: the same three-product arithmetic on pre-split planes)
    products = 3 if is_h2 else (X6_PRODUCTS if is_x6 else 1)
    peak = BF16_MFMA_PEAK_TFLOPS / products if (is_x6 or is_h2) else FP32_MFMA_PEAK_TFLOPS
    ach = kern[dom]["flop"] / (kern[dom]["ms"] * 1e-3) / 1e12
    r = {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
         "peak_basis": ("fp32-accurate split-fp16 kernel: dense fp16 MFMA peak 2500 TFLOP/s / 3 fp16 MFMAs per algorithmic MAC "
                        "block (csrc/conv_h2_kernel.h; csrc/conv_p2.hip for the VGG19 stack on pre-split planes); `achieved` counts "
                        "ALGORITHMIC fp32 FLOPs" if is_h2 else
                        "fp32-accurate split-bf16 kernel: dense bf16 MFMA peak 2500 TFLOP/s / 6 bf16 MFMAs per algorithmic MAC "
                        "block (csrc/conv_x6_kernel.h); `achieved` counts ALGORITHMIC fp32 FLOPs" if is_x6 else
                        "fp32-input MFMA (v_mfma_f32_32x32x2_f32) dense peak"),
         "achieved_vs_fp32_mfma_peak": ach / FP32_MFMA_PEAK_TFLOPS,
         "mfma_issued_tflops": ach * products,
         "traffic": None,
         "launches_per_step": kern[dom]["n"] // prof_steps,
         "avg_launch_us": 1e3 * kern[dom]["ms"] / kern[dom]["n"],
         "algorithmic_gflop_per_launch": kern[dom]["flop"] / kern[dom]["n"] / 1e9,
         "share_of_conv_time": kern[dom]["ms"] / tot_ms,
         "families": {k: {"ms_per_step": v["ms"] / prof_steps, "launches_per_step": v["n"] // prof_steps,
                          "tflops": v["flop"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0.0}
                      for k, v in fam.items()},
         "kernels": {k: {"ms_per_step": round(v["ms"] / prof_steps, 3), "avg_launch_us": round(1e3 * v["ms"] / v["n"], 1),
                         "tflops": round(v["flop"] / (v["ms"] * 1e-3) / 1e12, 1)}
                     for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["ms"])[:20]},
         "conv_ms_per_step": tot_ms / prof_steps,
         "whole_step_tflops": FLOP_PER_FRAME * batch / (1e-3 * ms_per_step) / 1e12}
    r["whole_step_vs_fp32_mfma_peak"] = r["whole_step_tflops"] / FP32_MFMA_PEAK_TFLOPS
    # the same denominators for the WHOLE step: algorithmic FLOPs of the step / step time against the split kernel's roof,
    # and the fp16 / bf16 MFMA FLOPs actually issued (products x algorithmic) against the dense 2.5 PFLOP/s peak
    r["whole_step_frac"] = r["whole_step_tflops"] / peak
    r["mfma_issued_frac_of_fp16_peak"] = ach * products / BF16_MFMA_PEAK_TFLOPS
    r["whole_step_mfma_issued_frac_of_fp16_peak"] = r["whole_step_tflops"] * products / BF16_MFMA_PEAK_TFLOPS
    # HBM traffic per launch of the dominant kernel: separate rocprofv3 PMC passes (tools/profile.sh), committed summary
    for rel in PMC_TRAFFIC:
        try:
            pmc = json.load(open(os.path.join(ROOT, rel)))
        except (OSError, ValueError):
            continue
        ent = pmc.get("kernels", {}).get(dom)
        if ent is not None:
            r["traffic"] = ent["hbm_bytes_per_launch"]
            r["traffic_source"] = {"file": rel, "collected_at_commit": pmc.get("head", ""), "correction": pmc.get("correction", "")}
        else:
            r["traffic_source"] = {"file": rel, "note": f"no entry for {dom}: traffic not reported (re-run tools/profile.sh)"}
        break
    return r


def behavior_row(vunet, device, size, frames=50, rows=16, iters=10):
    """BASELINE config 5, front half and end to end (experiments/behavior_net.py:1173-1184, data/data_conversions_3d.py:
    1130-1185): behaviour codes from the flow's reverse pass (config/behavior_net.yaml: 1024 channels, 2048 hidden, depth 2,
    15 blocks = 2.5 GB of fp32 weights streamed once per pass), the decoder's 50-step roll-out, then projection + raster +
    bf16 VunetAlter.transfer of one sequence.  Random weights (no checkpoint here).  Roofline of the flow pass: HBM, the
    weight bytes of its MLPs / 8 TB/s (csrc/seq.hip; profiles/r05_seq_time.txt)."""
    import numpy as np
    from behavior_driven_video_synthesis_amd.models.flow.simple_flow import UnsupervisedTransformer2
    from behavior_driven_video_synthesis_amd.models.pose_behavior_rnn import ResidualBehaviorNet
    from behavior_driven_video_synthesis_amd.render import PoseCamera, behavior_video
    torch.manual_seed(11)
    flow = UnsupervisedTransformer2(flow_in_channels=1024, flow_mid_channels=2048, flow_hidden_depth=2, n_flows=15)
    for blk in flow.flow.sub_layers:
        blk.norm_layer.initialized.fill_(1)
        for mlp in blk.coupling.s:     # an untrained scale net saturates its tanh: keep the pass conditioned like a trained one
            mlp.linears()[-1].weight.data.mul_(0.1)
    flow = flow.to(device)
    net = ResidualBehaviorNet(51, information_bottleneck=True, decoder_arch="lstm", linear_in_decoder=False, dim_hidden_b=1024)
    net.decoder.n_out.weight.data.mul_(0.05)
    net = net.to(device)
    z = torch.randn(rows, 1024, device=device)
    seq = 0.5 * torch.randn(rows, frames, 51, device=device)

    def timed(fn, batches=5):
        """Median over ``batches`` batches of ``iters`` calls (one host hiccup inside a 10 ms window otherwise triples a figure)."""
        for _ in range(3):
            fn()
        means = []
        for _ in range(batches):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                fn()
            torch.cuda.synchronize()
            means.append((time.perf_counter() - t0) / iters)
        return sorted(means)[len(means) // 2]
    with torch.no_grad():
        t_flow = timed(lambda: flow.reverse(z))
        b = flow.reverse(z).reshape(rows, 1024)
        t_dec = timed(lambda: net.generate_seq(b, seq, len=frames, start_frame=frames - 1))
        t_enc = timed(lambda: net.infer_b(seq, False))
    w_bytes = 4.0 * sum(p.numel() for n, p in flow.named_parameters() if ".main." in n and n.endswith("weight"))
    rng = np.random.RandomState(3)
    cam = PoseCamera((rng.randn(51) * 250.0).astype(np.float32), (40.0 + 40.0 * rng.rand(51)).astype(np.float32), list(range(51)),
                     np.concatenate([np.eye(3), np.array([[0.0], [0.0], [5000.0]])], axis=1), (1145.0, 500.0, 1145.0, 500.0),
                     (1000, 1000), size, device=device)
    app = (torch.rand(1, 3, size, size, device=device) * 2 - 1)
    was = vunet.training
    vunet.eval()
    t_e2e = timed(lambda: behavior_video(flow, net, vunet, app, seq[:1], frames, cam, z=z[:1], dtype="bf16", chunk=frames))
    vunet.train(was)
    ach = w_bytes / t_flow / 1e9
    # HBM bytes of one reverse pass: rocprofv3 PMC passes of tools/time_seq.py --only reverse (tools/profile.sh step 5), committed
    traffic = traffic_src = None
    for rel in PMC_TRAFFIC_SEQ:
        try:
            pmc = json.load(open(os.path.join(ROOT, rel)))
        except (OSError, ValueError):
            continue
        # (the pass's kernels: not the one-time packing of the tile-major weight copies)
        kern = {k: v for k, v in pmc.get("kernels", {}).items() if k.startswith("seq_") and not k.startswith("seq_pack")}
        coup = [v for k, v in kern.items() if k.startswith("seq_coupling_kernel")]
        if coup:
            passes = sum(v["launches_sampled"] for v in coup) / 31.0      # 31 coupling launches per 15-block reverse pass
            traffic = sum(v["hbm_bytes_per_launch"] * v["launches_sampled"] for v in kern.values()) / passes
            traffic_src = {"file": rel, "collected_at_commit": pmc.get("head", ""), "correction": pmc.get("correction", "")}
            break
    return {"workload": f"flow reverse ({rows} rows, 15 blocks, 1024 / 2048) + decoder roll-out ({frames} steps, 1024 hidden) + "
                        f"projection, raster and bf16 transfer of one {frames}-frame sequence at {size}x{size} (BASELINE configs[4])",
            "data": "synthetic, random weights",
            "flow_reverse_ms": 1e3 * t_flow, "decode_ms": 1e3 * t_dec, "encode_ms": 1e3 * t_enc, "rows": rows,
            "sequences_per_s_front_half": rows / (t_flow + t_dec),
            "end_to_end_ms_per_sequence": 1e3 * t_e2e, "end_to_end_frames_per_s": frames / t_e2e,
            "roofline": {"bound": "hbm", "kernel": "seq_linear_kernel (the flow's 120 MLP-layer launches)", "achieved": ach,
                         "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_pass": w_bytes,
                         "note": "whole reverse pass incl. its 31 coupling launches; per-CU L1 fill bound, see profiles/r05_seq_time.txt"}}


def behavior_train_row(device, rows=64, frames=50, iters=10):
    """BASELINE configs[3]: the two training stages of experiments/behavior_net.py at config/behavior_net.yaml's sizes, random
    weights, synthetic batches of 64 sequences.
      flow stage  (:703-714) UnsupervisedTransformer2 1024 / 2048 / depth 2 / 15 blocks = 629 M parameters: forward, FlowLoss,
                  backward, Adam(betas (0.5, 0.9)) with the weight gradient fused into the update (csrc/seq_train.hip), one
                  replayed hipGraph per step.  Roofline: HBM.  Algorithmic bytes per step = 28 B per weight (W read by the
                  forward pass; W, exp_avg, exp_avg_sq read and written once by the update) = 17.6 GB.
      cVAE stage  (:591-660) ResidualBehaviorNet (1024 hidden, 51 pose dimensions), 50 frames: forward, MSE + gamma KL,
                  back-propagation through time (100 cell steps), fused Adam, gamma controller (csrc/seq_bptt.hip)."""
    import copy
    from behavior_driven_video_synthesis_amd.experiments.behavior_net import BehaviorNet, DEFAULT_CONFIG
    torch.manual_seed(12)
    cfg = copy.deepcopy(DEFAULT_CONFIG)
    tr = BehaviorNet(cfg, n_kps=51, device=device)
    for blk in tr.latent_flow.flow.sub_layers:
        blk.norm_layer.initialized.fill_(1)
        for mlp in blk.coupling.s:     # an untrained scale net saturates its tanh: keep the pass conditioned like a trained one
            mlp.linears()[-1].weight.data.mul_(0.1)
    tr.net.decoder.n_out.weight.data.mul_(0.05)
    batch = {"keypoints": 0.5 * torch.randn(rows, frames + 1, 51, device=device)}

    def timed(fn, batches=5):
        """Median over ``batches`` batches of ``iters`` steps, HIP events around each batch."""
        for _ in range(3):
            fn()
        means = []
        for _ in range(batches):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(iters):
                fn()
            b.record()
            torch.cuda.synchronize()
            means.append(1e-3 * a.elapsed_time(b) / iters)
        return sorted(means)[len(means) // 2]
    t_vae = timed(lambda: tr.train_fn(batch, sync=False))
    out_vae = tr.train_fn(batch)
    bs = torch.randn(rows, 1024, device=device)
    noise = torch.randn(rows, 1024, device=device)
    t_flow = timed(lambda: tr.flow_engine.train_step(bs, noise))
    fl = tr.flow_engine.train_step(bs, noise).tolist()
    n_w = sum(p.numel() for n, p in tr.latent_flow.named_parameters() if ".main." in n and n.endswith("weight"))
    alg = 28.0 * n_w
    traffic = traffic_src = None
    for rel in PMC_TRAFFIC_SEQ_TRAIN:
        try:
            pmc = json.load(open(os.path.join(ROOT, rel)))
        except (OSError, ValueError):
            continue
        kern = {k: v for k, v in pmc.get("kernels", {}).items() if k.startswith("seq_")}
        ticks = [v for k, v in kern.items() if k.startswith("seq_flow_loss_kernel")]
        if ticks:   # one FlowLoss launch per step
            steps = sum(v["launches_sampled"] for v in ticks)
            traffic = sum(v["hbm_bytes_per_launch"] * v["launches_sampled"] for v in kern.values()) / steps
            traffic_src = {"file": rel, "collected_at_commit": pmc.get("head", ""), "correction": pmc.get("correction", ""),
                           "unit": "HBM bytes of all seq_* kernels per training step"}
            break
    row = {"workload": f"behavior_net training (BASELINE configs[3]), batch {rows}: flow stage = UnsupervisedTransformer2 1024 / 2048 / "
                       f"depth 2 / 15 blocks ({n_w / 1e6:.0f} M weights) forward + FlowLoss + backward + Adam; cVAE stage = "
                       f"ResidualBehaviorNet 1024 hidden / 51 dims over {frames} frames forward + MSE + gamma KL + BPTT + Adam",
           "data": "synthetic, random weights",
           "flow_stage": {"value": rows / t_flow, "unit": "samples/s", "ms_per_step": 1e3 * t_flow, "hip_graph": True,
                          "log": dict(zip(("flow_loss", "reference_nll_loss", "nlogdet_loss", "nll_loss"), fl)),
                          "roofline": {"bound": "hbm", "kernel": "the step's seq_* kernels (seq_dw_kernel: 15.1 of the 17.6 GB)",
                                       "achieved": alg / t_flow / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": alg / t_flow / 8e12,
                                       "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_step": alg,
                                       "note": "28 B per weight: W read by the forward pass; W, exp_avg, exp_avg_sq read and "
                                               "written once by the fused update.  The input-gradient chain reads W a second time "
                                               "(+4 B per weight of actual traffic)."}},
           "cvae_stage": {"value": rows / t_vae, "unit": "sequences/s", "ms_per_step": 1e3 * t_vae, "hip_graph": True,
                          "frames_per_s": rows * frames / t_vae,
                          "log": {k: out_vae[k] for k in ("loss", "loss_recon", "kl_loss", "gamma")}}}
    del tr
    torch.cuda.empty_cache()
    return row


def render_row(vunet, device, size, frames=50, chunk=50, iters=5):
    """BASELINE config 5 (the render half): a 50-frame pose sequence -> one raster launch -> batched VunetAlter.transfer
    (reference: per-frame cv2 raster + batch-1 transfer, data/data_conversions_3d.py:1130-1185).  Modes: fp32-accurate
    (split-fp16 kernels), and config 5's bf16 precision -- channel-blocked bf16 activations, fp32 accumulation
    (render_blk.BlockedTransfer) -- with one appearance encoding per sequence.  Roofline of its dominant kernel: HBM
    (algorithmic bytes / 8 TB/s); the layers sit between the HBM and the matrix-core bound, see DESIGN.md 5.3."""
    from behavior_driven_video_synthesis_amd import ops
    from behavior_driven_video_synthesis_amd.render import render_sequence
    g = torch.Generator().manual_seed(5)
    app = (torch.rand(1, 3, size, size, generator=g) * 2 - 1).to(device)
    kps = (torch.rand(frames, 17, 2, generator=g) * (size - 20) + 10).to(device)
    was = vunet.training
    vunet.eval()
    with torch.no_grad():
        eps = [torch.randn(m.shape, generator=g).to(device) for m in vunet.appearance_code(app)]

    def run(**kw):
        for _ in range(2):
            out, _ = render_sequence(vunet, app, kps, chunk=chunk, as_uint8=False, eps=eps, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            out, _ = render_sequence(vunet, app, kps, chunk=chunk, as_uint8=False, eps=eps, **kw)
        torch.cuda.synchronize()
        return out, (time.perf_counter() - t0) / iters
    ref, t32 = run()
    out, t16 = run(dtype="bf16", share_appearance=True)
    mse = float(((out - ref) ** 2).mean())
    peak = 2 * float(ref.abs().max())
    row = {"workload": f"{frames}-frame pose sequence -> GPU stickman raster -> batched VunetAlter.transfer at {size}x{size} "
                       "(BASELINE configs[4], render half)",
           "fp32_accurate": {"frames_per_s": frames / t32, "ms_per_sequence": 1e3 * t32},
           "bf16_shared_appearance": {"frames_per_s": frames / t16, "ms_per_sequence": 1e3 * t16,
                                      "psnr_vs_fp32_db": None if mse == 0 else 10 * float(torch.log10(torch.tensor(peak * peak / mse)))}}
    ops.profile_start()
    render_sequence(vunet, app, kps, chunk=chunk, as_uint8=False, eps=eps, dtype="bf16", share_appearance=True)
    fam = ops.profile_stop(detail=True)
    # dominant kernel of the bf16 render: the LDS-tiled convolution on channel-blocked bf16 activations (csrc/conv_blk.hip).
    # Algorithmic bytes per launch: bf16 sources + output (+ residual); the fp32 output layer writes 4 B per value.
    ms = nbytes = n_l = flop = 0.0
    for key, v in fam.items():
        if key[0] != "conv_blk_fwd" or key[-1] != "conv_blk_tiled_kernel":
            continue
        _, n, c1, c2, hs, ws, m, k, s_, has_res, nchw, _ = key
        ho, wo = (hs - 1) // s_ + 1, (ws - 1) // s_ + 1
        per = 2.0 * n * hs * ws * (c1 + c2) + (4.0 if nchw else 2.0) * n * ho * wo * m + (2.0 * n * ho * wo * m if has_res else 0.0)
        ms += v["ms"]
        nbytes += per * v["n"]
        n_l += v["n"]
        flop += 2.0 * n * ho * wo * m * (c1 + c2) * k * k * v["n"]
    if ms > 0:
        ach = nbytes / (ms * 1e-3) / 1e9
        row["roofline"] = {"bound": "hbm", "kernel": "conv_blk_tiled_kernel", "achieved": ach, "peak": 8000.0, "unit": "GB/s",
                           "frac": ach / 8000.0, "traffic": None, "algorithmic_bytes_per_launch": nbytes / n_l,
                           "avg_launch_us": 1e3 * ms / n_l,
                           "share_of_conv_time": ms / sum(v["ms"] for v in fam.values()),
                           # the same launches against the OTHER roof: these layers sit between the two
                           "mfma_achieved_tflops": flop / (ms * 1e-3) / 1e12,
                           "mfma_frac_of_bf16_peak": flop / (ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS}
        # mean HBM bytes per launch of that kernel over the sequence: rocprofv3 PMC passes of tools/bench_render.py
        # (tools/profile.sh step 4), committed summary
        for rel in PMC_TRAFFIC_RENDER:
            try:
                pmc = json.load(open(os.path.join(ROOT, rel)))
            except (OSError, ValueError):
                continue
            # (both forms of the tiled kernel: the wave-specialised one runs the 128-channel layers)
            ents = [v for k, v in pmc.get("kernels", {}).items()
                    if k.startswith("conv_blk_tiled_kernel") or k.startswith("conv_blk_ws_kernel")]
            if ents:   # every instantiation of the kernel, weighted by its launches
                nl = sum(e["launches_sampled"] for e in ents)
                row["roofline"]["traffic"] = sum(e["hbm_bytes_per_launch"] * e["launches_sampled"] for e in ents) / nl
                row["roofline"]["traffic_source"] = {"file": rel, "collected_at_commit": pmc.get("head", ""),
                                                     "launches_sampled": nl, "unit": "mean HBM bytes per launch"}
            break
    vunet.train(was)
    return row


def variant_rows(args, device, steps=10, warmup=3):
    """The optional terms of BASELINE config 2 and the pure-fp32 arithmetic, timed on THIS box beside the headline (same
    batch, same size, the step replayed from a captured hipGraph where the configuration can be recorded):
      regressor       the regressor side loop as the reference config runs it (experiments/shape_and_pose_net.py:407-425)
      gan             the PartDiscriminator adversarial term + one discriminator step (models/synth_discriminator.py:140-242;
                      the reference ships it but its loop never constructs it -- SURVEY F2)
      regressor+gan   both
      precision_f32   the headline step with every convolution on the fp32-input MFMA kernels (no operand split)."""
    import argparse
    import contextlib
    from behavior_driven_video_synthesis_amd import ops
    from behavior_driven_video_synthesis_amd.experiments.shape_and_pose_net import ShapePoseNet, synthetic_batch
    rows = {}
    for name, kw in (("regressor", {"regressor": True}), ("gan", {"gan": True}), ("regressor+gan", {"regressor": True, "gan": True}),
                     ("precision_f32", {"precision": "f32"})):
        a = argparse.Namespace(**{**vars(args), **kw})
        try:
            ops.set_conv_precision(a.precision)
            with contextlib.redirect_stdout(sys.stderr):
                tr = ShapePoseNet(make_config(a), device=device, total_steps=150000, vgg_synthetic=True, hip_graph=False)
                try:
                    tr.enable_hip_graph()
                except RuntimeError as e:
                    print(f"bench.py: variant {name}: hipGraph mode not available ({e}); issuing eagerly", file=sys.stderr)
            batch = synthetic_batch(a.batch, a.size, device, seed=42, with_regressor=a.regressor)
            settle = 0
            while tr._dev_sched and tr._capture and not tr._graphs and settle < 12:
                tr.train_fn(batch)
                settle += 1
            el, out = timed_steps(tr, batch, warmup, steps, torch.cuda.synchronize)
            rows[name] = {"value": a.batch * steps / el, "unit": "frames/s", "ms_per_step": 1e3 * el / steps, "steps": steps,
                          "warmup": warmup, "hip_graph": bool(tr._graphs), "conv_precision": a.precision,
                          "final_loss": float(out["loss"])}
            del tr, batch
        finally:
            ops.set_conv_precision(args.precision)
            ops.set_dropout_step(None)
    torch.cuda.empty_cache()
    return rows


def timed_steps(trainer, batch, warmup, steps, sync_all):
    for _ in range(warmup):
        trainer.train_fn(batch)
    sync_all()
    heartbeat("warm-up done")
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = trainer.train_fn(batch)
    timed_steps.host_issue_s = time.perf_counter() - t0     # the host's share: all launches issued, nothing awaited yet
    sync_all()
    return time.perf_counter() - t0, out


class GraphGuard:
    """In-process watchdog of the graph-replay phase of a multi-rank run (see main): armed before the first replay of a graph
    that holds RCCL collectives, cancelled when the timed replays have come back.  On expiry rank 0 writes the eager result it
    was given -- with the reason -- as THE JSON line, and the process leaves with os._exit (its stream is stuck in a collective:
    nothing can be torn down in order)."""

    def __init__(self, limit_s, json_fd, eager_result):
        import threading
        self._done = threading.Event()
        self._limit, self._fd, self._res = limit_s, json_fd, eager_result
        self._t = threading.Thread(target=self._run, daemon=True)
        self._t.start()

    def cancel(self):
        self._done.set()

    def _run(self):
        if self._done.wait(self._limit):
            return
        reason = f"the recorded graph (captured RCCL collectives) did not complete {self._limit:.0f} s after it was armed"
        print(f"bench.py: {reason}; reporting the eager measurement", file=sys.stderr, flush=True)
        if self._fd is not None:
            res = dict(self._res)
            res["config"] = dict(res["config"], hip_graph=False, fallback_reason=reason)
            os.write(self._fd, (json.dumps(res) + "\n").encode())
        os._exit(0)


def heartbeat(phase):
    """Rank 0 tells the launching parent that the run is alive (stderr; the parent's watchdog restarts a silent child tree)."""
    if os.environ.get("VUNET_BENCH_LAUNCHED") == "1" and int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench heartbeat] {phase}", file=sys.stderr, flush=True)


def run_with_watchdog(cmd, env, first_deadline, progress_deadline, echo=sys.stderr):
    """Run ``cmd`` as a child process TREE of its own (new session) and watch its stderr for ``[bench heartbeat]`` lines: no line
    within ``first_deadline`` seconds of the start, or within ``progress_deadline`` of the previous one, and the whole tree is
    killed (SIGKILL to the process group -- this process never touched the GPU, nothing is exec'ed).
    -> (return code or None if killed, stdout text, the last phase heard, reason or None)."""
    import signal
    import subprocess
    import threading
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    state = {"last": time.monotonic(), "phase": None, "beats": 0}
    out_chunks = []

    def pump_err():
        for line in p.stderr:
            if line.startswith("[bench heartbeat]"):
                state["last"], state["phase"] = time.monotonic(), line[len("[bench heartbeat]"):].strip()
                state["beats"] += 1
            if echo is not None:
                echo.write(line)
                echo.flush()

    def pump_out():
        for line in p.stdout:
            out_chunks.append(line)
    threads = [threading.Thread(target=pump_err, daemon=True), threading.Thread(target=pump_out, daemon=True)]
    for t in threads:
        t.start()
    reason = None
    while p.poll() is None:
        time.sleep(0.2)
        limit = first_deadline if state["beats"] == 0 else progress_deadline
        if time.monotonic() - state["last"] > limit:
            reason = (f"no heartbeat for {limit:.0f} s after " + (f"phase '{state['phase']}'" if state["beats"] else "the start"))
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            p.wait()
            break
    for t in threads:
        t.join(timeout=5)
    return (None if reason else p.returncode), "".join(out_chunks), state["phase"], reason


def launch_ranks(args):
    """``bench.py --gpus N`` outside a torchrun environment: start the N ranks as a fresh child process tree, relay rank
    0's JSON line.  Nothing in THIS process has touched the GPU (``torch.cuda.device_count()`` does not initialise HIP)
    and nothing will: the parent only waits -- with a watchdog.  Rank 0 prints a heartbeat after start-up, settle, warm-up
    and the timed steps; a child tree that goes silent (a replayed collective that never completes is the case this is
    for: with several ranks the step's RCCL all-reduces are captured into the hipGraph) is killed and ONE fresh tree is
    started with ``--hip-graph off``; the line then says so (``config.fallback_reason``).  A second failure exits
    non-zero.  Returns the exit code."""
    import socket
    n = args.gpus
    have = torch.cuda.device_count()
    if have < n:
        print(f"bench.py --gpus {n}: only {have} GPU(s) visible", file=sys.stderr)
        return 3
    first = float(os.environ.get("VUNET_BENCH_START_TIMEOUT", "600"))       # imports + RCCL bring-up on a fresh box
    progress = float(os.environ.get("VUNET_BENCH_PROGRESS_TIMEOUT", "240"))
    argv = list(sys.argv[1:])
    fallback_reason = None
    for attempt in range(2):
        with socket.socket() as s:                      # a free rendezvous port on the loopback interface
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL's intra-node transport needs it on this driver
        env["VUNET_BENCH_LAUNCHED"] = "1"
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
        print("bench.py: starting " + " ".join(cmd), file=sys.stderr)
        rc, out, phase, reason = run_with_watchdog(cmd, env, first, progress)
        line = None
        for ln in out.splitlines():
            if ln.startswith('{"metric"'):
                line = ln
            elif ln.strip():
                print(ln, file=sys.stderr)              # anything else the ranks wrote to stdout is not the result
        if reason is None and rc == 0 and line is not None:
            break
        why = reason or (f"the rank processes exited with code {rc}" if rc else "rank 0 printed no JSON line")
        print(f"bench.py --gpus {n}: attempt {attempt + 1} failed: {why}", file=sys.stderr)
        graph_off = "--hip-graph" in argv and argv[argv.index("--hip-graph") + 1] == "off"
        if attempt == 1 or graph_off:
            return rc if rc else 6
        # once more, a FRESH process tree, the step issued eagerly (no captured collectives)
        if "--hip-graph" in argv:
            argv[argv.index("--hip-graph") + 1] = "off"
        else:
            argv += ["--hip-graph", "off"]
        fallback_reason = f"first attempt (--hip-graph auto): {why}"
    res = json.loads(line)
    if res.get("n_gpus") != n or res.get("config", {}).get("rccl_world_size") != n:
        print(f"bench.py --gpus {n}: the line reports n_gpus={res.get('n_gpus')}, "
              f"rccl_world_size={res.get('config', {}).get('rccl_world_size')}", file=sys.stderr)
        return 5
    res.setdefault("config", {})["fallback_reason"] = fallback_reason
    sys.stdout.write(json.dumps(res) + "\n")
    sys.stdout.flush()
    return 0


def main():
    args = parse()
    if args.cpu_child:
        return cpu_baseline_child(args.cpu_child)
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("VUNET_DP_FORCE") == "1"):
        raise SystemExit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started as one rank of {world}: the two must agree")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries ONE JSON line.  Native libraries write there too (RCCL prints a version banner through C stdio when
    # its first communicator comes up): keep the real stdout aside for the JSON line and point fd 1 at stderr meanwhile.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("VUNET_DP_FORCE") == "1":
        dist.init_process_group("nccl", device_id=device)
    heartbeat("process group up")
    from behavior_driven_video_synthesis_amd import ops
    from behavior_driven_video_synthesis_amd.experiments.shape_and_pose_net import ShapePoseNet, synthetic_batch
    ops.set_conv_precision(args.precision)
    ops.apply_env_tuning()   # VUNET_TUNING=key=value,...: A/B runs of dispatcher choices (none set: the library's own)

    cfg = make_config(args)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):  # the constructors print like the reference does; stdout is for the JSON line
        trainer = ShapePoseNet(cfg, device=device, total_steps=150000, vgg_synthetic=True, hip_graph=False)
        # auto: replay wherever the configuration can be captured.  With several ranks that needs the step's all-reduces on
        # the C-ABI RCCL communicator (ordinary stream operations: capturable); an eagerly issued step is 12-22 ms of
        # Python on a slow-host box, exactly what a scaling figure is sensitive to.  A capture that fails on ANY rank sends
        # every rank back to eager issue together (ShapePoseNet._capture_agreed; said on stderr, `config.hip_graph` false).
        want_graph = args.hip_graph == "on" or (args.hip_graph == "auto" and (not trainer.averager.active or trainer.averager.native))
        # Several ranks, `auto`: the K timed steps are measured EAGERLY first (a complete, valid result), then again from the
        # recorded graph under a watchdog.  Replaying captured RCCL collectives on N ranks has never run on hardware: if the
        # replay does not come back, rank 0 prints the eager line (config.fallback_reason says why) and every rank leaves
        # with os._exit -- the run cannot end without a number, whoever launched the ranks (bench.py's own launcher adds a
        # second net: launch_ranks' heartbeat watchdog and one restart with --hip-graph off).
        two_phase = want_graph and args.hip_graph == "auto" and trainer.averager.active
        if want_graph and not two_phase:
            try:
                trainer.enable_hip_graph()
            except RuntimeError as e:      # a configuration the capture does not cover (stated by the trainer)
                if args.hip_graph == "on":
                    raise
                print(f"bench.py: hipGraph mode not available here ({e}); issuing eagerly", file=sys.stderr)
    batch = synthetic_batch(args.batch, args.size, device, seed=42, with_regressor=args.regressor, rank=rank)
    heartbeat("trainer built")

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def build_result(elapsed, out, settle, host_issue_ms, fallback_reason=None):
        loss_val = float(out["loss"])
        assert loss_val == loss_val, "loss is NaN"
        res = {
            "metric": "frames/sec VUnet 256x256 bs=16 fwd+bwd", "value": world * args.batch * args.steps / elapsed,
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"h2": "f32 (operands scaled by a power of two and split into 2 fp16 terms, 3 partial products on fp16 MFMA, "
                            "fp32 accumulate; weight gradients and uncovered layers: 3 bf16 terms / 6 products, fp32-input "
                            "MFMA, fp32 VALU)",
                      "x6": "f32 (operands split exactly into 3 bf16 terms, 6 partial products on bf16 MFMA, fp32 accumulate; "
                            "fp32-input MFMA / fp32 VALU for the layers the split kernels do not cover)",
                      "f32": "f32"}[args.precision],
            "data": "synthetic",
            "config": {"workload": f"Human3.6m shape_and_pose_net VunetAlter {args.size}x{args.size} per-GPU bs={args.batch} "
                                   "fwd+bwd, VGG19 perceptual + KL loss, fused Adam, dropout 0.05"
                                   + (", regressor side loop on" if args.regressor else ", regressor side loop off")
                                   + (", adversarial term on" if args.gan else "")
                                   + ", seeded-synthetic VGG19 weights, rasterised synthetic stickmen",
                       "global_batch": world * args.batch, "parallelism": f"dp{world}", "conv_precision": args.precision,
                       "flop_per_frame": FLOP_PER_FRAME, "final_loss": loss_val,
                       "hip_streams": 1 if trainer.vunet._side_stream is None else 4,
                       "hip_graph": bool(trainer._graphs), "graph_settle_steps": settle,
                       "host_issue_ms_per_step": host_issue_ms,
                       "rccl_world_size": dist.get_world_size() if dist.is_initialized() else 1,
                       "dp_backend": trainer.averager.backend,
                       "allreduce_ms_per_step": trainer.averager.mean_allreduce_ms(),
                       # share of the all-reduce time that ran while backward was still computing (HIP events)
                       "allreduce_overlap_frac": trainer.averager.overlap_fraction(),
                       # (HIP events cannot be read back from a captured graph: under replay the two figures above are those of the
                       #  eagerly issued steps before the capture -- same kernels, same streams)
                       "allreduce_timed_on": ("eager steps before the capture" if trainer._graphs else "timed steps")
                                             if trainer.averager.active else None},
        }
        res["config"]["fallback_reason"] = fallback_reason
        return res

    eager_first, guard = None, None
    if two_phase:
        for _ in range(6):                 # past the initialisation batches: the step has its final form
            trainer.train_fn(batch)
        el, out_e = timed_steps(trainer, batch, args.warmup, args.steps, sync_all)
        t = torch.tensor([el], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        eager_first = build_result(float(t.item()), out_e, 0, None)
        heartbeat("eager timed steps done")
        limit = float(os.environ.get("VUNET_BENCH_GRAPH_TIMEOUT", "180"))
        guard = GraphGuard(limit, json_fd if rank == 0 else None, eager_first)
        try:
            trainer.enable_hip_graph()
        except RuntimeError as e:
            print(f"bench.py: hipGraph mode not available here ({e}); the eager measurement stands", file=sys.stderr)
            guard.cancel()
            guard = None

    # graph mode: the step is recorded after the initialisation batches (the KL term joins the loss then) and two eager
    # steps of the final form -- all of that happens HERE, before the W warm-up steps, so that the timed region holds
    # K replays and nothing else
    settle = 0
    while trainer._dev_sched and trainer._capture and not trainer._graphs and settle < 12:
        trainer.train_fn(batch)
        settle += 1
    sync_all()
    heartbeat(f"settled ({settle} steps, graph {'recorded' if trainer._graphs else 'not in use'})")
    elapsed, out = timed_steps(trainer, batch, args.warmup, args.steps, sync_all)
    if guard is not None:
        guard.cancel()                     # the replayed steps came back: the graph measurement is the result
    heartbeat("timed steps done")
    # host time to ISSUE one step: measured on steps that start with an empty device queue (inside the timed loop the host
    # runs ahead until the launch queue is full and is then throttled to the GPU's pace, which says nothing about the host)
    issue = []
    for _ in range(3):
        sync_all()
        t0 = time.perf_counter()
        trainer.train_fn(batch)
        issue.append(time.perf_counter() - t0)
    sync_all()
    host_issue_ms = 1e3 * statistics.median(issue)
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    result = build_result(elapsed, out, settle, host_issue_ms)
    if eager_first is not None:
        result["config"]["eager_issue"] = {"value": eager_first["value"], "ms_per_step": eager_first["ms_per_step"],
                                           "note": "the same K timed steps issued eagerly, measured first (see --hip-graph)"}

    if result["config"]["rccl_world_size"] != args.gpus:
        raise SystemExit(f"rank {rank}: RCCL communicator spans {result['config']['rccl_world_size']} rank(s), --gpus {args.gpus}")
    if dist.is_initialized():
        # self-validation of the data-parallel run: after the timed steps every rank must hold bit-identical parameters
        # (bucket checksums all-gathered and compared); a number from diverged replicas is not a result
        ok = trainer.averager.replicas_consistent()
        result["config"]["dp_consistent"] = ok
        if not ok:
            raise SystemExit(f"rank {rank}: replicas diverged after {args.warmup + args.steps} steps (bucket checksums differ)")

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
        capture, trainer._capture = getattr(trainer, "_capture", False), False   # event pairs need eager launches
        sync_all()
        ops.profile_start()
        for _ in range(prof_steps):
            trainer.train_fn(batch)
        sync_all()
        trainer.vunet._side_stream = side_stream
        trainer._capture = capture
        ops.enable_wgrad_streams(wgrad_streams)
    heartbeat("instrumented steps done")
    if rank == 0 and not args.no_roofline:
        recs = ops._prof["recs"][:]                       # raw (key, flop, ev0, ev1) records of the instrumented steps
        fam = ops.profile_stop()                           # per family: conv_gather_fwd / conv_gather_dgrad / conv_wgrad
        ops._prof["recs"] = recs
        kern = ops.profile_stop(by_kernel=True)            # per kernel instantiation, rocprofv3 spelling
        tot_ms = sum(v["ms"] for v in fam.values())
        dom = max(kern, key=lambda k: kern[k]["ms"])
        result["roofline"] = roofline_entry(kern, fam, dom, tot_ms, prof_steps, result["ms_per_step"], args.batch)
    elif not args.no_roofline:
        ops.profile_stop()

    if (rank == 0 and world == 1 and not dist.is_initialized() and not args.no_variants
            and not (args.regressor or args.gan) and args.precision == "h2"):
        result["variants"] = variant_rows(args, device)
    if rank == 0 and world == 1 and not args.no_render and args.size % 32 == 0:
        result["render"] = render_row(trainer.vunet, device, args.size)
        result["behavior"] = behavior_row(trainer.vunet, device, args.size)
        result["behavior_train"] = behavior_train_row(device)
    # BASELINE config 1 (Market 128^2, bs 2, 30-channel 64x64 appearance input): plumbing rows, GPU and CPU
    cfg1 = batch1 = None
    if rank == 0 and world == 1 and not args.no_config1:
        cfg1 = market_config()
        with contextlib.redirect_stdout(sys.stderr):
            tr1 = ShapePoseNet(cfg1, device=device, n_channels_x=30, total_steps=150000, vgg_synthetic=True)
        batch1 = synthetic_batch(2, 128, device, seed=42, n_channels_x=30, appearance_size=64)
        el1, out1 = timed_steps(tr1, batch1, 5, 20, torch.cuda.synchronize)
        result["config1"] = {"workload": "Market1501 shape_and_pose_net VunetAlter 128x128 bs=2, x = 30x64x64 (BASELINE configs[0])",
                             "value": 2 * 20 / el1, "unit": "frames/s", "ms_per_step": 1e3 * el1 / 20,
                             "host_issue_ms_per_step": 1e3 * timed_steps.host_issue_s / 20,
                             "final_loss": float(out1["loss"])}
        # the same trainer, the step replayed from a captured hipGraph (the host issues one launch instead of ~1900)
        if not tr1.averager.active:
            tr1.enable_hip_graph()
            el1g, out1g = timed_steps(tr1, batch1, 5, 20, torch.cuda.synchronize)
            result["config1"]["hip_graph"] = {"value": 2 * 20 / el1g, "unit": "frames/s", "ms_per_step": 1e3 * el1g / 20,
                                              "host_issue_ms_per_step": 1e3 * timed_steps.host_issue_s / 20,
                                              "captured": bool(tr1._graphs), "final_loss": float(out1g["loss"])}
            ops.set_dropout_step(None)
        del tr1
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args, cfg, batch, cfg1, batch1)
    if rank == 0:
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    os.close(json_fd)
    if dist.is_initialized():
        from behavior_driven_video_synthesis_amd.parallel import shutdown_native_comm
        dist.barrier()
        shutdown_native_comm()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
