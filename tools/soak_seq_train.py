#!/usr/bin/env python3
"""Soak of config 4's training on the MI355X path: does it LEARN, and does it stay put?

Flow stage at config/behavior_net.yaml's sizes (1024 / 2048 / depth 2 / 15 blocks, batch 64, fresh flow: ActNorm initialises itself
from the first batch): SOAK_ITERS fused steps (replayed graph) on codes drawn from a fixed correlated Gaussian mixture -- the
negative log-likelihood the step reports must fall below the reference nll of a standard normal draw + a margin it starts far above,
every parameter stays finite, device memory stays flat.  cVAE stage (1024 hidden, 51 dims, 50 frames, batch 64) on smooth synthetic
pose sequences: the reconstruction error must fall.  Prints one JSON line.

    python tools/soak_seq_train.py           (SOAK_ITERS, default 1500)
"""
import copy
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from behavior_driven_video_synthesis_amd.experiments.behavior_net import BehaviorNet, DEFAULT_CONFIG
    iters = int(os.environ.get("SOAK_ITERS", "1500"))
    torch.manual_seed(0)
    cfg = copy.deepcopy(DEFAULT_CONFIG)
    cfg["training"].update(flow_lr=2e-6, lr_init=1e-3)      # (faster than the reference's schedule: the soak is short)
    tr = BehaviorNet(cfg, n_kps=51)
    dev = tr.device
    g = torch.Generator(device=dev).manual_seed(1)
    # behaviour codes: a two-component mixture with a low-rank correlation (what a flow has to undo)
    mix = 0.3 * torch.randn(1024, 64, device=dev, generator=g)
    centres = 1.5 * torch.randn(2, 1024, device=dev, generator=g)

    def codes(n):
        k = torch.randint(0, 2, (n,), device=dev, generator=g)
        return centres[k] + torch.randn(n, 64, device=dev, generator=g) @ mix.t() + 0.5 * torch.randn(n, 1024, device=dev, generator=g)
    t0 = time.time()
    flow_log, mem, mem_v = [], [], []
    eng = tr.flow_engine
    for it in range(iters):
        sc = eng.train_step(codes(64))
        if it % 100 == 0 or it == iters - 1:
            v = sc.tolist()
            assert all(math.isfinite(x) for x in v), (it, v)
            flow_log.append((it, round(v[0] / 1024, 4), round(v[3] / 1024, 4), round(v[2] / 1024, 4)))
            mem.append(round(torch.cuda.memory_allocated() / 2 ** 20, 1))
    t_flow = time.time() - t0
    assert all(torch.isfinite(p).all() for p in tr.latent_flow.parameters())
    # per-dimension loss in nats: a standard normal has 0.5 (+ 0.92 of the constant FlowLoss leaves out); the data's own entropy bounds it
    first, last = flow_log[0][1], flow_log[-1][1]
    # the cVAE stage on smooth sequences: sums of a few sinusoids per pose dimension
    tt = torch.linspace(0, 1, 51, device=dev)[None, :, None]

    def poses(n):
        a = torch.randn(n, 1, 51, 3, device=dev, generator=g)
        f = torch.rand(n, 1, 51, 3, device=dev, generator=g) * 6.0
        ph = torch.rand(n, 1, 51, 3, device=dev, generator=g) * 6.28
        return 0.3 * (a * torch.sin(f * tt[..., None] * 6.28 + ph)).sum(-1)
    t0 = time.time()
    vae_log = []
    for it in range(iters):
        out = tr.train_fn({"keypoints": poses(64)}, sync=(it % 100 == 0 or it == iters - 1))
        if it % 100 == 0 or it == iters - 1:
            assert math.isfinite(out["loss"]), (it, out)
            vae_log.append((it, round(out["loss_recon"], 5), round(out["kl_loss"], 3), round(out["gamma"], 6)))
            mem_v.append(round(torch.cuda.memory_allocated() / 2 ** 20, 1))
    t_vae = time.time() - t0
    assert all(torch.isfinite(p).all() for p in tr.net.parameters())
    print(json.dumps({"iterations": iters, "flow_stage": {"seconds": round(t_flow, 1), "ms_per_step_incl_data": round(1e3 * t_flow / iters, 2),
                                                          "loss_per_dim (it, total, nll, -logdet)": flow_log},
                      "cvae_stage": {"seconds": round(t_vae, 1), "ms_per_step_incl_data": round(1e3 * t_vae / iters, 2),
                                     "(it, recon, kl, gamma)": vae_log},
                      "allocated_MiB_flow_stage": mem, "allocated_MiB_cvae_stage": mem_v}))
    assert last < first - 0.2, f"the flow's loss did not fall: {first} -> {last}"
    assert vae_log[-1][1] < 0.5 * vae_log[0][1], f"the cVAE's reconstruction error did not fall: {vae_log[0]} -> {vae_log[-1]}"
    for m in (mem, mem_v):      # flat once the stage's plan exists (the first sample is taken after the first step)
        assert max(m) <= 1.02 * m[1] + 1.0, "device memory keeps growing"


if __name__ == "__main__":
    main()
