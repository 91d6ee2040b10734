#!/usr/bin/env python3
"""Soak of the behaviour front half: many flow / roll-out calls with changing batch sizes and lengths; device memory must stay flat
once the per-shape plans exist (recordings and their buffers are capped per engine, seq._GraphCache.MAX_GRAPHS)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


@torch.no_grad()
def main():
    from behavior_driven_video_synthesis_amd.models.flow.simple_flow import UnsupervisedTransformer2
    from behavior_driven_video_synthesis_amd.models.pose_behavior_rnn import ResidualBehaviorNet
    iters = int(os.environ.get("SOAK_ITERS", "1500"))
    torch.manual_seed(0)
    flow = UnsupervisedTransformer2(flow_in_channels=256, flow_mid_channels=512, flow_hidden_depth=2, n_flows=4)
    for blk in flow.flow.sub_layers:
        blk.norm_layer.initialized.fill_(1)
    flow = flow.cuda()
    net = ResidualBehaviorNet(51, information_bottleneck=True, decoder_arch="lstm", dim_hidden_b=256).cuda()
    g = torch.Generator().manual_seed(1)
    peak, t0, checks = [], time.time(), []
    for it in range(iters):
        rows = int(torch.randint(1, 97, (1,), generator=g))
        length = int(torch.randint(1, 40, (1,), generator=g))
        t_in = int(torch.randint(1, 6, (1,), generator=g))
        z = torch.randn(rows, 256, device="cuda")
        b = flow.reverse(z).reshape(rows, 256)
        back, _ = flow(b)
        x = 0.5 * torch.randn(rows, t_in, 51, device="cuda")
        xs, *_ = net.generate_seq(b, x, len=length, start_frame=-1)
        net.infer_b(xs, False)
        if it % 100 == 99:
            torch.cuda.synchronize()
            err = float((back.reshape(rows, 256) - z).abs().max())
            checks.append(err)
            peak.append(torch.cuda.memory_allocated() / 2 ** 20)
            assert torch.isfinite(xs).all()
    print(json.dumps({"iterations": iters, "seconds": round(time.time() - t0, 1), "allocated_MiB_every_100": [round(p, 1) for p in peak],
                      "max_round_trip_error": max(checks), "recordings_flow": len(flow.flow.engine().graph.graphs),
                      "recordings_net": len(net.engine().graph.graphs)}))
    assert max(peak[len(peak) // 2:]) <= 1.05 * max(peak[:len(peak) // 2]) + 1.0, "device memory keeps growing"


if __name__ == "__main__":
    main()
