#!/usr/bin/env python3
"""Time the streaming 1x1 convolution kernel on the bs-16 step's `nin` shapes (needs an MI355X):  python tools/time_1x1.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from behavior_driven_video_synthesis_amd import ops  # noqa: E402

# (name, n, c, h, w, mode, in_act / aux_act)
SHAPES = [("32ch 256^2 fwd elu", 16, 32, 256, 256, 0, 1), ("32ch 256^2 dgrad elu'", 16, 32, 256, 256, 1, 1),
          ("64ch 128^2 fwd elu", 16, 64, 128, 128, 0, 1), ("64ch 128^2 dgrad elu'", 16, 64, 128, 128, 1, 1),
          ("128ch 64^2 fwd elu", 16, 128, 64, 64, 0, 1), ("128ch 64^2 dgrad elu'", 16, 128, 64, 64, 1, 1),
          ("128ch 32^2 fwd elu", 16, 128, 32, 32, 0, 1), ("128ch 32^2 dgrad elu'", 16, 128, 32, 32, 1, 1),
          ("128ch 16^2 fwd elu", 16, 128, 16, 16, 0, 1)]
# the 3-channel layers (VGG19 conv1_1, the pyramids' first convolutions, the output layer's data gradient): fp32 FMA kernel
for name, n, cin, cout, h, w, mode in [("3 -> 64 3x3 256^2 fwd + relu", 16, 3, 64, 256, 256, 0),
                                      ("3 -> 32 3x3 256^2 fwd", 16, 3, 32, 256, 256, 0),
                                      ("3 -> 32 3x3 256^2 dgrad (dy 3 ch)", 16, 3, 32, 256, 256, 1),
                                      ("32 -> 3 3x3 256^2 fwd (out_conv)", 16, 32, 3, 256, 256, 0),
                                      ("64 -> 3 3x3 256^2 dgrad (conv1_1)", 16, 64, 3, 256, 256, 1)]:
    g = torch.Generator().manual_seed(1)
    ci, co = (cin, cout) if mode == 0 else (cout, cin)       # the convolution's own channels; the launch reads cin
    v = (torch.randn(co, ci, 3, 3, generator=g) * 0.1).cuda()
    wt_f, wt_d, scale, shift, invnorm, wx_f, wx_d = ops.pack_weights(v, None, None, None, None, ci, 0, 1, True)
    x = torch.randn(n, cin, h, w, generator=g).cuda()
    y = torch.empty(n, cout, h, w, device="cuda")
    d = ops.ConvDesc(N=n, C1=cin, C2=0, Hs=h, Ws=w, M=cout, m_off=0, Mpad=(wt_f if mode == 0 else wt_d).shape[1], Ho=h, Wo=w,
                     KH=3, KW=3, stride=1, pad=1, mode=mode, in_act=0, in_slope=0.0, drop_p=0.0, drop_seed=0,
                     out_act=2 if "relu" in name else 0, d2s=0, aux_act=0, aux_slope=0.0, aux_drop_p=0.0, aux_drop_seed=0)

    def launch3():
        ops._call("vunet_conv2d_gather", ctypes.byref(d), ops._p(x), None, ops._p(wt_f if mode == 0 else wt_d),
                  ops._p(shift) if mode == 0 else None, None, None, ops._p(y), ops._stream())
    for _ in range(3):
        launch3()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        launch3()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name:36s} {us:7.1f} us   {(x.numel() + y.numel()) * 4 / 1e9 / (us * 1e-6) / 1e3:5.2f} TB/s   "
          f"{2.0 * n * h * w * cin * cout * 9 / (us * 1e-6) / 1e12:5.1f} TF/s")

for name, n, c, h, w, mode, act in SHAPES:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, c, h, w, generator=g).cuda()
    aux = torch.randn(n, c, h, w, generator=g).cuda() if mode == 1 else None
    v = (torch.randn(c, c, 1, 1, generator=g) * 0.1).cuda()
    wt_f, wt_d, scale, shift, invnorm, wx_f, wx_d = ops.pack_weights(v, None, None, None, None, c, 0, 1, True)
    y = torch.empty(n, c, h, w, device="cuda")
    d = ops.ConvDesc(N=n, C1=c, C2=0, Hs=h, Ws=w, M=c, m_off=0, Mpad=(wt_f if mode == 0 else wt_d).shape[1], Ho=h, Wo=w,
                     KH=1, KW=1, stride=1, pad=0, mode=mode, in_act=act if mode == 0 else 0, in_slope=0.0, drop_p=0.0,
                     drop_seed=0, out_act=0, d2s=0, aux_act=act if mode == 1 else 0, aux_slope=0.0, aux_drop_p=0.0,
                     aux_drop_seed=0)

    def launch():
        ops._call("vunet_conv2d_gather", ctypes.byref(d), ops._p(x), None, ops._p(wt_f if mode == 0 else wt_d),
                  ops._p(shift) if mode == 0 else None, None, ops._p(aux), ops._p(y), ops._stream())
    for _ in range(3):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        launch()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    gb = (2 + (mode == 1)) * x.numel() * 4 / 1e9
    print(f"{name:24s} {us:7.1f} us   {gb / (us * 1e-6) / 1e3:5.2f} TB/s")
