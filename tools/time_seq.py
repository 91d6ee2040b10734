#!/usr/bin/env python3
"""Time the behaviour front half of config 5 at the reference configuration (config/behavior_net.yaml): the flow's reverse
pass (1024 channels, 2048 hidden, depth 2, 15 blocks: 2.5 GB of fp32 weights) on 16 rows, and the 50-step decoder roll-out
(dim_hidden_b 1024, 51 pose dimensions).  HIP events around replayed graphs; weights are random (no checkpoint here).

    python tools/time_seq.py [--rows 16] [--flows 15] [--reps 20] [--eager]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=16)
    ap.add_argument("--flows", type=int, default=15)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--only", default=None, choices=[None, "reverse"], help="time the flow's reverse pass only (counter runs)")
    args = ap.parse_args()
    from behavior_driven_video_synthesis_amd.models.flow.simple_flow import UnsupervisedTransformer2
    from behavior_driven_video_synthesis_amd.models.pose_behavior_rnn import ResidualBehaviorNet
    torch.manual_seed(0)
    flow = UnsupervisedTransformer2(flow_in_channels=1024, flow_mid_channels=2048, flow_hidden_depth=2, n_flows=args.flows)
    for blk in flow.flow.sub_layers:
        blk.norm_layer.initialized.fill_(1)
        for net in blk.coupling.s:
            net.linears()[-1].weight.data.mul_(0.1)
    flow = flow.cuda()
    net = ResidualBehaviorNet(51, information_bottleneck=True, decoder_arch="lstm", linear_in_decoder=False, dim_hidden_b=1024).cuda()
    if args.eager:
        flow.flow.engine().graph.enabled = False
        net.engine().graph.enabled = False
    z = torch.randn(args.rows, 1024, device="cuda")
    x = 0.5 * torch.randn(args.rows, 50, 51, device="cuda")
    w_bytes = sum(p.numel() for n, p in flow.named_parameters() if ".main." in n and n.endswith("weight")) * 4
    res = {"rows": args.rows, "flows": args.flows, "graph": not args.eager}
    t = timed(lambda: flow.flow.engine()._run(z, True), args.reps)
    res["flow_reverse_ms"] = round(t, 4)
    res["flow_weight_GB"] = round(w_bytes / 1e9, 3)
    res["flow_weight_stream_GBps"] = round(w_bytes / t / 1e6, 1)
    if args.only == "reverse":
        print(json.dumps(res))
        return
    t = timed(lambda: flow.flow.engine()._run(z, False), args.reps)
    res["flow_forward_ms"] = round(t, 4)
    b = flow.reverse(z).reshape(args.rows, 1024)
    t = timed(lambda: net.generate_seq(b, x, len=50, start_frame=49), args.reps)
    res["decode_50_ms"] = round(t, 4)
    gate_bytes = 4096 * (64 + 1024) * 4
    res["decode_gate_stream_GBps"] = round(50 * gate_bytes / t / 1e6, 1)
    t = timed(lambda: net.infer_b(x, False), args.reps)
    res["encode_50_ms"] = round(t, 4)
    print(json.dumps(res))


if __name__ == "__main__":
    with torch.no_grad():
        main()
