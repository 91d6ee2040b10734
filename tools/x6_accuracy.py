#!/usr/bin/env python3
"""Accuracy study of the split-bf16 convolution (csrc/conv_x6_kernel.h) against fp64, beside the fp32-MFMA kernel.

    python tools/x6_accuracy.py            (needs an MI355X)

Prints, per problem, max and rms error relative to the largest output for: the x6 kernel, the fp32-MFMA kernel,
and the x6 kernel on operands that are exactly representable in bf16 (then only the matrix core's own fp32
accumulation can contribute error)."""
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from behavior_driven_video_synthesis_amd import ops  # noqa: E402


def run(x, v, use_x6, nt):
    n, c1, h, w = x.shape
    cout = v.shape[0]
    wt_f, wt_d, scale, shift, invnorm, wx_f, wx_d = ops.pack_weights(v, None, None, None, None, c1, 0, 1, False)
    y = torch.empty(n, cout, h, w, device="cuda")
    d = ops.ConvDesc(N=n, C1=c1, C2=0, Hs=h, Ws=w, M=cout, m_off=0, Mpad=wt_f.shape[1], Ho=h, Wo=w, KH=3, KW=3,
                     stride=1, pad=1, mode=0, in_act=0, in_slope=0.0, drop_p=0.0, drop_seed=0, out_act=0, d2s=0)
    os.environ["VUNET_X6_FORCE_NT"] = str(nt)
    if use_x6:
        ops._call("vunet_conv2d_x6", ctypes.byref(d), ops._p(x), None, ops._p(wx_f), None, None, None, None, ops._p(y),
                  ops._stream())
    else:
        ops._call("vunet_conv2d_gather", ctypes.byref(d), ops._p(x), None, ops._p(wt_f), None, None, None, ops._p(y),
                  ops._stream())
    torch.cuda.synchronize()
    return y


def err(y, ref):
    e = (y.double().cpu() - ref)
    s = float(ref.abs().max())
    return float(e.abs().max()) / s, float(e.pow(2).mean().sqrt()) / s


for cin, cout, nt in [(16, 32, 1), (32, 32, 4), (48, 32, 2), (64, 64, 2), (64, 32, 2), (128, 64, 2), (256, 128, 2), (512, 64, 2)]:
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(2, cin, 16, 32, generator=g).cuda()
    v = (torch.randn(cout, cin, 3, 3, generator=g) * 0.1).cuda()
    ref = F.conv2d(x.double().cpu(), v.double().cpu(), padding=1)
    e6, e32 = err(run(x, v, True, nt), ref), err(run(x, v, False, nt), ref)
    xb, vb = x.bfloat16().float(), v.bfloat16().float()
    refb = F.conv2d(xb.double().cpu(), vb.double().cpu(), padding=1)
    e6b, e32b = err(run(xb, vb, True, nt), refb), err(run(xb, vb, False, nt), refb)
    print(f"cin {cin:4d} cout {cout:4d} K {9 * cin:5d} | x6 max {e6[0]:.2e} rms {e6[1]:.2e} | f32 max {e32[0]:.2e} rms {e32[1]:.2e}"
          f" | bf16-exact operands: x6 max {e6b[0]:.2e} rms {e6b[1]:.2e}, f32 max {e32b[0]:.2e} rms {e32b[1]:.2e}")
