#!/usr/bin/env python3
"""Accuracy study of the split convolutions against fp64, beside the fp32-MFMA kernel.

    python tools/x6_accuracy.py            (needs an MI355X)

Per problem: max and rms error relative to the largest output for
  h2   two scaled fp16 terms, three partial products (csrc/conv_h2_kernel.h)
  x6   three bf16 terms, six partial products        (csrc/conv_x6_kernel.h)
  f32  the fp32-input MFMA kernel (an fmaf chain)
first on Gaussian operands, then on inputs whose magnitude varies over 2^-24 .. 1 from pixel column to pixel column
(error then measured per output column, relative to that column's own largest output: the case a tensor-wide scale
could hurt) and on operands that are exactly representable in bf16 (only the matrix core's own fp32 accumulation can
contribute error)."""
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from behavior_driven_video_synthesis_amd import ops  # noqa: E402


def run(x, v, mode, nt):
    n, c1, h, w = x.shape
    cout = v.shape[0]
    ops.set_conv_precision(mode)
    wt_f, wt_d, scale, shift, invnorm, wx_f, wx_d = ops.pack_weights(v, None, None, None, None, c1, 0, 1, False)
    y = torch.empty(n, cout, h, w, device="cuda")
    d = ops.ConvDesc(N=n, C1=c1, C2=0, Hs=h, Ws=w, M=cout, m_off=0, Mpad=wt_f.shape[1], Ho=h, Wo=w, KH=3, KW=3,
                     stride=1, pad=1, mode=0, in_act=0, in_slope=0.0, drop_p=0.0, drop_seed=0, out_act=0, d2s=0)
    ops.set_tuning("split_force_nt", nt)
    if mode != "f32":
        amax = ops.absmax_partials(x) if mode == "h2" else None
        ops._call("vunet_conv2d_x6", ctypes.byref(d), ops._p(x), None, ops._p(wx_f), None, None, None, None, ops._p(y),
                  ops._p(amax), None, ops._stream())
    else:
        ops._call("vunet_conv2d_gather", ctypes.byref(d), ops._p(x), None, ops._p(wt_f), None, None, None, ops._p(y),
                  ops._stream())
    torch.cuda.synchronize()
    return y


def err(y, ref, per_column=False):
    e = (y.double().cpu() - ref)
    s = ref.abs().amax(dim=(0, 1, 2), keepdim=True) if per_column else ref.abs().max()
    r = e / s
    return float(r.abs().max()), float(r.pow(2).mean().sqrt())


def fmt(tag, e):
    return f"{tag} max {e[0]:.2e} rms {e[1]:.2e}"


for cin, cout, nt in [(16, 32, 1), (32, 32, 4), (48, 32, 2), (64, 64, 2), (64, 32, 2), (128, 64, 2), (256, 128, 2), (512, 64, 2)]:
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(2, cin, 16, 32, generator=g).cuda()
    v = (torch.randn(cout, cin, 3, 3, generator=g) * 0.1).cuda()
    ref = F.conv2d(x.double().cpu(), v.double().cpu(), padding=1)
    line = [f"cin {cin:4d} cout {cout:4d} K {9 * cin:5d}"]
    line.append(" ".join(fmt(m, err(run(x, v, m, nt), ref)) for m in ("h2", "x6", "f32")))
    # magnitude 2^-(3c/4) in pixel column c (0 .. 2^-23.25), constant over the 3-column receptive field up to 2^-1.5
    col = torch.pow(2.0, -0.75 * torch.arange(32, dtype=torch.float32)).view(1, 1, 1, 32).cuda()
    xw = x * col
    refw = F.conv2d(xw.double().cpu(), v.double().cpu(), padding=1)
    line.append("wide range, per column: " + " ".join(fmt(m, err(run(xw, v, m, nt), refw, True)) for m in ("h2", "x6", "f32")))
    xb, vb = x.bfloat16().float(), v.bfloat16().float()
    refb = F.conv2d(xb.double().cpu(), vb.double().cpu(), padding=1)
    line.append("bf16-exact operands: " + " ".join(fmt(m, err(run(xb, vb, m, nt), refb)) for m in ("h2", "x6", "f32")))
    print(" | ".join(line))
