#!/usr/bin/env python3
"""Time the direct weight-gradient kernel (stride-2, 1x1 and small-map layers of the bs-16 step; needs an MI355X):
python tools/time_wgrad_direct.py     (VUNET_HIP_LIB=<variant> for A/B builds; 20 back-to-back launches per shape)"""
import ctypes
import sys

import torch

sys.path.insert(0, "/root/repo")
from behavior_driven_video_synthesis_amd import ops  # noqa: E402

ops.set_conv_precision("h2")
SHAPES = [  # name, N, C1, C2, Cout, H, W, k, stride, in_act, drop
    ("s2 32->64 256^2", 16, 32, 0, 64, 256, 256, 3, 2, 0, 0.0),
    ("s2 64->128 128^2", 16, 64, 0, 128, 128, 128, 3, 2, 0, 0.0),
    ("s2 128->128 64^2", 16, 128, 0, 128, 64, 64, 3, 2, 0, 0.0),
    ("1x1 nin 32->32 256^2 elu", 16, 32, 0, 32, 256, 256, 1, 1, 1, 0.0),
    ("1x1 nin 64->32 256^2 elu", 16, 64, 0, 32, 256, 256, 1, 1, 1, 0.0),
    ("1x1 nin 128->64 128^2 elu", 16, 128, 0, 64, 128, 128, 1, 1, 1, 0.0),
    ("3x3 128+128->128 16^2 elu+drop", 16, 128, 128, 128, 16, 16, 3, 1, 1, 0.05),
    ("3x3 128+128->128 8^2 elu+drop", 16, 128, 128, 128, 8, 8, 3, 1, 1, 0.05),
    ("3x3 128->128 4^2 elu+drop", 16, 128, 0, 128, 4, 4, 3, 1, 1, 0.05),
]
for name, n, c1, c2, cout, h, w, k, s, act, drop in SHAPES:
    ho, wo = (h - 1) // s + 1 if k == 3 else h, (w - 1) // s + 1 if k == 3 else w
    x1 = torch.randn(n, c1, h, w, device="cuda")
    x2 = torch.randn(n, c2, h, w, device="cuda") if c2 else None
    dy = torch.randn(n, cout, ho, wo, device="cuda")
    wd = ops.WgradDesc(N=n, C1=c1, C2=c2, Hs=h, Ws=w, Cout=cout, Ho=ho, Wo=wo, KH=k, KW=k, stride=s, pad=1 if k == 3 else 0, in_act=act,
                       in_slope=0.0, drop_p=drop, drop_seed=123, nsplit=1, flags=2)
    ns = ops._lib.lib().vunet_conv2d_wgrad_nsplit(ctypes.byref(wd))
    wd.nsplit = ns
    ktot = k * k * (c1 + c2)
    slabs = torch.empty(ns * ops._r32(cout) * (ktot + 1), device="cuda")
    dshift = slabs[ns * ops._r32(cout) * ktot:]
    ax, ad = ops.absmax_partials(x1, x2), ops.absmax_partials(dy)
    buf = ctypes.create_string_buffer(96)
    ops._call("vunet_conv2d_wgrad_variant", ctypes.byref(wd), buf, 96)

    def launch():
        ops._call("vunet_conv2d_wgrad", ctypes.byref(wd), ops._p(x1), ops._p(x2), ops._p(dy), ops._p(slabs), ops._p(dshift), ops._p(ax),
                  ops._p(ad), ops._stream())
    for _ in range(3):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        launch()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name:34s} {us:8.1f} us {2.0 * n * ho * wo * (c1 + c2) * cout * k * k / (us * 1e-6) / 1e12:6.1f} TF/s  nsplit {ns:4d}  {buf.value.decode()}")
