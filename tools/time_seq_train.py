#!/usr/bin/env python3
"""Time the flow stage's training step of BASELINE config 4 at the reference configuration (config/behavior_net.yaml: 1024
channels, 2048 hidden, depth 2, 15 blocks = 629 M parameters; batch 64): forward + FlowLoss + backward + Adam fused into the
weight-gradient sweep, replayed from a hipGraph.  HIP events around the replays; random weights.

    python tools/time_seq_train.py [--rows 64] [--flows 15] [--reps 10] [--eager]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=64)
    ap.add_argument("--flows", type=int, default=15)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--stage", default="flow", choices=["flow", "cvae"])
    args = ap.parse_args()
    if args.stage == "cvae":
        return cvae(args)
    from behavior_driven_video_synthesis_amd.models.flow.simple_flow import UnsupervisedTransformer2
    torch.manual_seed(0)
    flow = UnsupervisedTransformer2(flow_in_channels=1024, flow_mid_channels=2048, flow_hidden_depth=2, n_flows=args.flows)
    for blk in flow.flow.sub_layers:
        blk.norm_layer.initialized.fill_(1)
        for net in blk.coupling.s:
            net.linears()[-1].weight.data.mul_(0.1)
    flow = flow.cuda()
    eng = flow.flow.train_engine(lr=4.5e-7 * 64, betas=(0.5, 0.9), weight_decay=0.0)
    eng.graph.enabled = not args.eager
    x = torch.randn(args.rows, 1024, device="cuda")
    noise = torch.randn(args.rows, 1024, device="cuda")
    for _ in range(3):
        out = eng.train_step(x, noise)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(args.reps):
        out = eng.train_step(x, noise)
    b.record()
    torch.cuda.synchronize()
    t = a.elapsed_time(b) / args.reps
    n_w = sum(p.numel() for n, p in flow.named_parameters() if ".main." in n and n.endswith("weight"))
    res = {"rows": args.rows, "flows": args.flows, "graph": not args.eager, "step_ms": round(t, 4), "samples_per_s": round(1e3 * args.rows / t, 1),
           "weights_M": round(n_w / 1e6, 1),
           # algorithmic bytes: W read by the forward pass; W, exp_avg, exp_avg_sq read and written by the update = 7 x 4 B / weight
           "algorithmic_GB": round(28.0 * n_w / 1e9, 3), "achieved_GBps": round(28.0 * n_w / t / 1e6, 1),
           "loss": [round(v, 4) for v in out.tolist()]}
    print(json.dumps(res))


def cvae(args):
    """The first stage's step at config/behavior_net.yaml's sizes: ResidualBehaviorNet 1024 hidden / 51 dims, 50 frames."""
    import copy
    from behavior_driven_video_synthesis_amd.experiments.behavior_net import BehaviorNet, DEFAULT_CONFIG
    cfg = copy.deepcopy(DEFAULT_CONFIG)
    cfg["architecture"].update(n_flows=1, flow_mid_channels_factor=1, flow_hidden_depth=1)
    torch.manual_seed(0)
    tr = BehaviorNet(cfg, n_kps=51, hip_graph=not args.eager)
    tr.net.decoder.n_out.weight.data.mul_(0.05)
    batch = {"keypoints": 0.5 * torch.randn(args.rows, 51, 51, device="cuda")}
    for _ in range(3):
        tr.train_fn(batch, sync=False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(args.reps):
        tr.train_fn(batch, sync=False)
    b.record()
    torch.cuda.synchronize()
    t = a.elapsed_time(b) / args.reps
    out = tr.train_fn(batch)
    print(json.dumps({"stage": "cvae", "rows": args.rows, "frames": 50, "graph": not args.eager, "step_ms": round(t, 4),
                      "sequences_per_s": round(1e3 * args.rows / t, 1), "loss": round(out["loss"], 5), "kl": round(out["kl_loss"], 4)}))


if __name__ == "__main__":
    main()
