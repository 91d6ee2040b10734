#!/usr/bin/env python3
"""Time the split convolution kernels on the layer shapes that dominate the bs-16 256x256 step (needs an MI355X).

    python tools/time_conv.py [--modes h2,x6] [--nt 0]      (--nt: ops.set_tuning("split_force_nt"), 0 = the dispatcher's choice)

Per shape and scheme: average launch time over 20 launches (HIP events on the launch stream), algorithmic TFLOP/s."""
import argparse
import ctypes
import os
import sys

os.environ.setdefault("VUNET_ALLOW_TIMING_BUILD", "1")   # A/B libraries of tools/ab_build.sh may be timing-only builds

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from behavior_driven_video_synthesis_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--modes", default="h2,x6")
ap.add_argument("--nt", type=int, default=0)
args = ap.parse_args()
if args.nt:
    ops.set_tuning("split_force_nt", args.nt)

# (name, n, cin, cout, h, w, mode, in_act, masked)
SHAPES = [
    ("vgg conv2_2 fwd", 16, 128, 128, 128, 128, 0, 0, False),
    ("vgg conv1_2 fwd", 16, 64, 64, 256, 256, 0, 0, False),
    ("vgg conv3_x fwd", 16, 256, 256, 64, 64, 0, 0, False),
    ("vgg conv2_2 dgrad+relu", 16, 128, 128, 128, 128, 1, 0, True),
    ("vunet rnb 64ch fwd elu", 16, 64, 64, 128, 128, 0, 1, False),
    ("vunet 32ch 256^2 fwd elu", 16, 32, 32, 256, 256, 0, 1, False),
    ("vunet up 64->128 d2s 128^2", 16, 64, 128, 128, 128, 0, 1, "d2s"),
    ("vunet up 128->256 d2s 64^2", 16, 128, 256, 64, 64, 0, 1, "d2s"),
    ("vunet 32ch 256^2 dgrad", 16, 32, 32, 256, 256, 1, 0, False),
    ("vunet 32ch 256^2 dgrad elu' drop", 16, 32, 32, 256, 256, 1, 0, "elu+drop"),
    ("vunet 128ch 64^2 dgrad", 16, 128, 128, 64, 64, 1, 0, False),
    ("vunet 64ch 128^2 dgrad elu'", 16, 64, 64, 128, 128, 1, 0, "elu"),
    ("vunet 64ch 128^2 dgrad elu' drop", 16, 64, 64, 128, 128, 1, 0, "elu+drop"),
    ("vunet 128ch 32^2 fwd elu", 16, 128, 128, 32, 32, 0, 1, False),
    ("vunet 128ch 32^2 dgrad elu'", 16, 128, 128, 32, 32, 1, 0, "elu"),
    ("vgg conv4_x fwd", 16, 512, 512, 32, 32, 0, 0, False),
    ("vgg conv2_2 fwd, N = 32", 32, 128, 128, 128, 128, 0, 0, False),
    ("K = 4608, 64^2", 16, 512, 128, 64, 64, 0, 0, False),
    ("vgg conv5_x fwd (16 wide)", 16, 512, 512, 16, 16, 0, 0, False),
    ("vgg conv5_x dgrad+relu", 16, 512, 512, 16, 16, 1, 0, True),
]
for name, n, cin, cout, h, w, mode, in_act, masked in SHAPES:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, cin, h, w, generator=g).cuda()
    v = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).cuda()
    d2s = masked == "d2s"                              # the up-convolution's sub-pixel store
    masked = False if d2s else masked
    auxk = masked if isinstance(masked, str) else ""   # data gradient * ELU'(aux) (+ the forward pass's dropout mask)
    masked = bool(masked) and not auxk
    m = torch.randn(n, cin, h, w, generator=g).cuda() if masked else None
    aux = torch.randn(n, cin, h, w, generator=g).cuda() if auxk else None
    line = [f"{name:28s}"]
    for sch in args.modes.split(","):
        ops.set_conv_precision(sch)
        wt_f, wt_d, scale, shift, invnorm, wx_f, wx_d = ops.pack_weights(v, None, None, None, None, cin, 0, 1, True)
        mo = cout if mode == 0 else cin
        y = torch.empty(n, mo, h, w, device="cuda")
        d = ops.ConvDesc(N=n, C1=cin if mode == 0 else cout, C2=0, Hs=h, Ws=w, M=mo, m_off=0,
                         Mpad=(wt_f if mode == 0 else wt_d).shape[1], Ho=h, Wo=w, KH=3, KW=3, stride=1, pad=1, mode=mode,
                         in_act=in_act, in_slope=0.0, drop_p=0.0, drop_seed=0, out_act=0, d2s=int(d2s),
                         aux_act=1 if auxk else 0, aux_slope=0.0, aux_drop_p=0.05 if "drop" in auxk else 0.0, aux_drop_seed=7)
        wx = wx_f if mode == 0 else wx_d
        amax = ops.absmax_partials(x) if sch == "h2" else None

        if w % 32 and sch != "h2":
            line.append(f"{sch} (not covered)")
            continue

        def launch():
            ops._call("vunet_conv2d_x6", ctypes.byref(d), ops._p(x), None, ops._p(wx), None, None, ops._p(aux), ops._p(m), ops._p(y),
                      ops._p(amax), None, ops._stream())
        for _ in range(3):
            launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            launch()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        tf = 2.0 * n * h * w * cin * cout * 9 / (us * 1e-6) / 1e12
        buf = ctypes.create_string_buffer(96)
        ops._call("vunet_conv2d_variant", ctypes.byref(d), 0, {"x6": 1, "h2": 2}[sch], int(masked), buf, 96)
        line.append(f"{sch} {us:8.1f} us {tf:6.1f} TF/s {buf.value.decode().replace('conv_', '').replace('_kernel', '')}")
        if sch == "h2":
            e0.record()
            for _ in range(20):
                ops.absmax_partials(x)
            e1.record()
            torch.cuda.synchronize()
            line.append(f"(absmax {e0.elapsed_time(e1) / 20 * 1e3:6.1f} us)")
    print(" | ".join(line))
