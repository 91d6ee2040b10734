#!/usr/bin/env python3
"""Time the split weight-gradient kernel (h2 scheme) on the bs-16 layer shapes (needs an MI355X):  python tools/time_wgrad.py"""
import ctypes, os, sys, torch
sys.path.insert(0, "/root/repo")
from behavior_driven_video_synthesis_amd import ops
ops.set_conv_precision("h2")
for name, n, c1, cout, h, w in [("128ch 128^2", 16, 128, 128, 128, 128), ("64ch 256^2", 16, 64, 64, 256, 256), ("256ch 64^2", 16, 256, 256, 64, 64), ("32ch 256^2", 16, 32, 32, 256, 256), ("64+64->64 128^2", 16, 128, 64, 128, 128),
                               ("32->3 out_conv 256^2 (fp32 FMA kernel)", 16, 32, 3, 256, 256)]:
    x = torch.randn(n, c1, h, w, device="cuda"); dy = torch.randn(n, cout, h, w, device="cuda")
    wd = ops.WgradDesc(N=n, C1=c1, C2=0, Hs=h, Ws=w, Cout=cout, Ho=h, Wo=w, KH=3, KW=3, stride=1, pad=1, in_act=1, in_slope=0.0, drop_p=0.0, drop_seed=0, nsplit=1, flags=2)
    ns = ops._lib.lib().vunet_conv2d_wgrad_nsplit(ctypes.byref(wd)); wd.nsplit = ns
    ktot = 9 * c1
    slabs = torch.empty(ns * ops._r32(cout) * (ktot + 1), device="cuda"); dshift = slabs[ns * ops._r32(cout) * ktot:]
    ax, ad = ops.absmax_partials(x), ops.absmax_partials(dy)
    def launch():
        ops._call("vunet_conv2d_wgrad", ctypes.byref(wd), ops._p(x), None, ops._p(dy), ops._p(slabs), ops._p(dshift), ops._p(ax), ops._p(ad), ops._stream())
    for _ in range(3): launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): launch()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name:18s} {us:8.1f} us {2.0*n*h*w*c1*cout*9/(us*1e-6)/1e12:6.1f} TF/s nsplit {ns}")
