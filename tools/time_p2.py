#!/usr/bin/env python3
"""Time vunet_p2_conv (pre-split planes, csrc/conv_p2.hip) beside the h2 kernel of vunet_conv2d on the VGG19 layer shapes of
the bs-16 256x256 step (needs an MI355X).

    python tools/time_p2.py [--form 0|1|2] [--reps 20]

Per shape: average launch time over --reps back-to-back launches (HIP events; sustained, clock-limited regime), algorithmic
TFLOP/s and fraction of the 833 TFLOP/s three-product roof -- forward (ReLU epilogue) and data gradient (masked)."""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from behavior_driven_video_synthesis_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--form", type=int, default=0)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--batch", type=int, default=16)
args = ap.parse_args()
ops.set_tuning("p2_form", args.form)

SHAPES = [("conv1_2", 64, 64, 256), ("conv2_1", 64, 128, 128), ("conv2_2", 128, 128, 128), ("conv3_1", 128, 256, 64),
          ("conv3_x", 256, 256, 64), ("conv4_1", 256, 512, 32), ("conv4_x", 512, 512, 32), ("conv5_x", 512, 512, 16)]


def timed(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.reps * 1e3


n = args.batch
print(f"batch {n}, p2 form {args.form} (0 = dispatcher), {args.reps} back-to-back launches per figure")
tot = {"p2f": 0.0, "h2f": 0.0, "p2d": 0.0, "h2d": 0.0}
for name, cin, cout, s in SHAPES:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, cin, s, s, generator=g).clamp_min(0).cuda()
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5).cuda()
    bias = (torch.randn(cout, generator=g) * 0.1).cuda()
    dy = (torch.randn(n, cout, s, s, generator=g) * 1e-3).cuda()
    flop = 2.0 * n * s * s * cin * cout * 9
    # ---- p2
    W = ops.P2Weights(wt, bias)
    px, pdy = ops.p2_from_nchw(x), ops.p2_from_nchw(dy)
    out, gout = ops.Planes((n, cout, s, s), "cuda"), ops.Planes((n, cin, s, s), "cuda")
    W.image(False), W.image(True)
    us_f = timed(lambda: ops.p2_conv(px, W, out))
    us_d = timed(lambda: ops.p2_conv(pdy, W, gout, dgrad=True, mask=px))
    # ---- h2 (fp32 NCHW in and out; the |x| maxima from a tag, as in the step)
    wt_f, wt_d, scale, shift, invnorm, wx_f, wx_d = ops.pack_weights(wt, None, bias, None, None, cin, 0, 1, True)
    y = torch.empty(n, cout, s, s, device="cuda")
    dx = torch.empty(n, cin, s, s, device="cuda")
    df = ops.ConvDesc(N=n, C1=cin, C2=0, Hs=s, Ws=s, M=cout, m_off=0, Mpad=wt_f.shape[1], Ho=s, Wo=s, KH=3, KW=3, stride=1, pad=1,
                      mode=0, in_act=0, in_slope=0.0, drop_p=0.0, drop_seed=0, out_act=ops.ACT_RELU, d2s=0)
    dd = ops.ConvDesc(N=n, C1=cout, C2=0, Hs=s, Ws=s, M=cin, m_off=0, Mpad=wt_d.shape[1], Ho=s, Wo=s, KH=3, KW=3, stride=1, pad=1,
                      mode=1, in_act=0, in_slope=0.0, drop_p=0.0, drop_seed=0, out_act=0, d2s=0, aux_act=ops.ACT_RELU)
    ax, ady = ops.absmax_partials(x), ops.absmax_partials(dy)
    am = torch.zeros(1024, device="cuda")
    us_hf = timed(lambda: ops._call("vunet_conv2d_a2", ctypes.byref(df), ops._p(x), None, ops._p(wt_f), ops._p(wx_f), ops._p(shift),
                                    None, None, None, ops._p(y), ops._p(ax), None, ops._p(am), ops._stream()))
    us_hd = timed(lambda: ops._call("vunet_conv2d_a2", ctypes.byref(dd), ops._p(dy), None, ops._p(wt_d), ops._p(wx_d), None, None,
                                    None, ops._p(x), ops._p(dx), ops._p(ady), None, ops._p(am), ops._stream()))
    tot["p2f"] += us_f; tot["h2f"] += us_hf; tot["p2d"] += us_d; tot["h2d"] += us_hd
    f = lambda us: f"{us:7.1f} us {flop / us / 1e6:6.1f} TF/s {flop / us / 1e6 / 833.3:5.3f}"
    print(f"{name:8s} {cin:3d}->{cout:3d} @{s:3d}^2 | fwd  p2 {f(us_f)} | h2 {f(us_hf)} | dgrad p2 {f(us_d)} | h2 {f(us_hd)}")
print("sum over the listed shapes (us): " + ", ".join(f"{k} {v:.0f}" for k, v in tot.items()))
