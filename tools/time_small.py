#!/usr/bin/env python3
"""Back-to-back timing of the small-map layers (csrc/conv_h2_small.hip) on the bs-16 shapes of the VUnet bottleneck.

    python tools/time_small.py [--reps 200]

Per shape: the dispatcher's own kernel (h2 scheme) and the fp32 gather kernel it replaced, us per launch over `reps`
launches issued back to back on one stream (launch overhead included: it is what the step pays).
"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from behavior_driven_video_synthesis_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=200)
args = ap.parse_args()

# (name, n, c1, c2, cout, hs, stride, mode, in_act, drop)
SHAPES = [
    ("rnb 4x4 fwd elu+drop", 16, 128, 0, 128, 4, 1, 0, 1, 0.05),
    ("rnb 4x4 fwd dual", 16, 128, 128, 128, 4, 1, 0, 1, 0.05),
    ("rnb 8x8 fwd elu+drop", 16, 128, 0, 128, 8, 1, 0, 1, 0.05),
    ("rnb 8x8 fwd dual", 16, 128, 128, 128, 8, 1, 0, 1, 0.05),
    ("rnb 16x16 fwd elu+drop", 16, 128, 0, 128, 16, 1, 0, 1, 0.05),
    ("rnb 16x16 fwd dual", 16, 128, 128, 128, 16, 1, 0, 1, 0.05),
    ("up 4x4 128->512", 16, 128, 0, 512, 4, 1, 0, 0, 0.0),
    ("up 8x8 128->512", 16, 128, 0, 512, 8, 1, 0, 0, 0.0),
    ("down 8->4", 16, 128, 0, 128, 8, 2, 0, 0, 0.0),
    ("down 16->8", 16, 128, 0, 128, 16, 2, 0, 0, 0.0),
    ("down 32->16", 16, 128, 0, 128, 32, 2, 0, 0, 0.0),
    ("dgrad 4x4", 16, 128, 0, 128, 4, 1, 1, 0, 0.0),
    ("dgrad 8x8", 16, 128, 0, 128, 8, 1, 1, 0, 0.0),
    ("dgrad 16x16", 16, 128, 0, 128, 16, 1, 1, 0, 0.0),
    ("dgrad 8x8 from 512", 16, 512, 0, 128, 8, 1, 1, 0, 0.0),
    ("dgrad s2 4->8", 16, 128, 0, 128, 4, 2, 1, 0, 0.0),
    ("dgrad s2 8->16", 16, 128, 0, 128, 8, 2, 1, 0, 0.0),
    ("dgrad s2 16->32", 16, 128, 0, 128, 16, 2, 1, 0, 0.0),
]
ops.set_conv_precision("h2")
for name, n, c1, c2, cout, hs, stride, mode, in_act, drop in SHAPES:
    g = torch.Generator().manual_seed(1)
    x1 = torch.randn(n, c1, hs, hs, generator=g).cuda()
    x2 = torch.randn(n, c2, hs, hs, generator=g).cuda() if c2 else None
    if mode == 0:
        v = (torch.randn(cout, c1 + c2, 3, 3, generator=g) * 0.05).cuda()
        ho = hs if stride == 1 else hs // 2
        wt_f, wt_d, scale, shift, invnorm, wx_f, wx_d = ops.pack_weights(v, None, None, None, None, c1, c2, 1, False)
        wt, wx = wt_f, wx_f
    else:   # data gradient: x1 is dy (c1 = forward Cout), cout = forward Cin
        v = (torch.randn(c1, cout, 3, 3, generator=g) * 0.05).cuda()
        ho = hs * stride
        wt_f, wt_d, scale, shift, invnorm, wx_f, wx_d = ops.pack_weights(v, None, None, None, None, cout, 0, 1, True)
        wt, wx = wt_d, wx_d
    y = torch.empty(n, cout, ho, ho, device="cuda")
    d = ops.ConvDesc(N=n, C1=c1, C2=c2, Hs=hs, Ws=hs, M=cout, m_off=0, Mpad=wt.shape[1], Ho=ho, Wo=ho, KH=3, KW=3,
                     stride=stride, pad=1, mode=mode, in_act=in_act, in_slope=0.0, drop_p=drop, drop_seed=7, out_act=0, d2s=0)
    amax = ops.absmax_partials(x1, x2)
    flop = 2.0 * n * (ho * ho if mode == 0 else hs * hs) * (c1 + c2) * cout * 9
    line = [f"{name:24s}"]
    for which in ("h2", "f32"):
        def launch():
            ops._call("vunet_conv2d", ctypes.byref(d), ops._p(x1), ops._p(x2), ops._p(wt), ops._p(wx) if which == "h2" else None,
                      None, None, None, ops._p(y), ops._p(amax) if which == "h2" else None, None, ops._stream())
        for _ in range(5):
            launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.reps):
            launch()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / args.reps * 1e3
        buf = ctypes.create_string_buffer(96)
        ops._call("vunet_conv2d_variant", ctypes.byref(d), 0, 2 if which == "h2" else 0, 0, buf, 96)
        line.append(f"{which} {us:7.1f} us {flop / (us * 1e-6) / 1e12:6.1f} TF/s {buf.value.decode().replace('conv_', '').replace('_kernel', ''):28s}")
    print(" | ".join(line))
