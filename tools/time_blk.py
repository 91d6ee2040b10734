#!/usr/bin/env python3
"""Per-layer timing of the blocked-bf16 render convolutions (csrc/conv_blk.hip) at the render loop's shapes
(VunetAlter.transfer_code, 25 frames of 256x256, DEFAULT_CONFIG widths), under each value of the tile-height knob.

    python tools/time_blk.py [--iters 20] [--frames 25]
One line per layer: microseconds, algorithmic GB/s (bf16 sources + output + residual) and TFLOP/s.
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from behavior_driven_video_synthesis_amd import ops  # noqa: E402
from behavior_driven_video_synthesis_amd.render_blk import blk_empty  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--frames", type=int, default=25)
ap.add_argument("--uniform", action="store_true", help="the uniform form of the tiled kernel everywhere (blk_ws = 1)")
ap.add_argument("--ws", action="store_true", help="the wave-specialised form everywhere (blk_ws = 2)")
ap.add_argument("--global-weights", action="store_true", help="the direct kernel with weights from global memory (blk_ws = 3)")
ap.add_argument("--tiled-only", action="store_true", help="only the layers the LDS-tiled kernel runs (A/B builds)")
a = ap.parse_args()
N = a.frames

# (C1, C2, H, M, k, stride, elu, res, d2s, nchw)
LAYERS = [
    (32, 0, 256, 32, 3, 1, 1, 1, 0, 0), (32, 32, 256, 32, 3, 1, 1, 1, 0, 0), (32, 0, 256, 3, 3, 1, 0, 0, 0, 1),
    (32, 0, 256, 32, 1, 1, 1, 0, 0, 0), (32, 0, 256, 64, 3, 2, 0, 0, 0, 0),
    (64, 0, 128, 64, 3, 1, 1, 1, 0, 0), (64, 64, 128, 64, 3, 1, 1, 1, 0, 0), (64, 0, 128, 128, 3, 1, 0, 0, 1, 0),
    (64, 0, 128, 64, 1, 1, 1, 0, 0, 0), (64, 0, 128, 128, 3, 2, 0, 0, 0, 0),
    (128, 0, 64, 128, 3, 1, 1, 1, 0, 0), (128, 128, 64, 128, 3, 1, 1, 1, 0, 0), (128, 0, 64, 256, 3, 1, 0, 0, 1, 0),
    (128, 0, 64, 128, 1, 1, 1, 0, 0, 0), (128, 0, 64, 128, 3, 2, 0, 0, 0, 0),
    (128, 0, 32, 128, 3, 1, 1, 1, 0, 0), (128, 128, 32, 128, 3, 1, 1, 1, 0, 0), (128, 0, 32, 512, 3, 1, 0, 0, 1, 0),
    (128, 128, 16, 128, 3, 1, 1, 1, 0, 0), (128, 0, 16, 512, 3, 1, 0, 0, 1, 0), (128, 128, 8, 128, 3, 1, 1, 1, 0, 0),
    (128, 128, 4, 128, 3, 1, 1, 1, 0, 0), (128, 0, 4, 128, 1, 1, 1, 0, 0, 0),
]


def run(layer, nt):
    c1, c2, h, m, k, s, elu, res, d2s, nchw = layer
    pad = k // 2
    ho = (h + 2 * pad - k) // s + 1
    g = torch.Generator(device="cuda").manual_seed(1)
    x1 = torch.randn(N, c1 // 8, h, h, 8, device="cuda", generator=g).to(torch.bfloat16)
    x2 = torch.randn(N, c2 // 8, h, h, 8, device="cuda", generator=g).to(torch.bfloat16) if c2 else None
    mpad = (m + 31) // 32 * 32
    wb = (torch.randn((c1 + c2) * k * k * mpad, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    shift = torch.zeros(m, device="cuda")
    if nchw:
        y = torch.empty(N, m, ho, ho, device="cuda")
    elif d2s:
        y = blk_empty(N, m // 4, 2 * ho, 2 * ho, "cuda")
    else:
        y = blk_empty(N, m, ho, ho, "cuda")
    r = x1 if res else None
    d = ops.ConvDesc(N=N, C1=c1, C2=c2, Hs=h, Ws=h, M=m, m_off=0, Mpad=mpad, Ho=ho, Wo=ho, KH=k, KW=k, stride=s, pad=pad,
                     mode=0, in_act=ops.ACT_ELU if elu else ops.ACT_NONE, in_slope=0.0, drop_p=0.0, drop_seed=0,
                     out_act=ops.ACT_NONE, d2s=d2s)
    ops.set_tuning("blk_force_nt", nt)
    ops.set_tuning("blk_ws", 1 if a.uniform else (2 if a.ws else (3 if a.global_weights else 0)))

    def call():
        ops._call("vunet_conv2d_blk", ctypes.byref(d), ops._p(x1), ops._p(x2), ops._p(wb), ops._p(shift), ops._p(r), ops._p(y),
                  nchw, ops._stream())
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    ops.set_tuning("blk_force_nt", 0)
    us = 1e3 * e0.elapsed_time(e1) / a.iters
    nbytes = 2.0 * N * h * h * (c1 + c2) + (4.0 if nchw else 2.0) * N * ho * ho * m + (2.0 * N * ho * ho * m if res else 0.0)
    flop = 2.0 * N * ho * ho * m * (c1 + c2) * k * k
    return us, nbytes / us / 1e3, flop / us / 1e6


for layer in LAYERS:
    if a.tiled_only and not (layer[4] == 3 and layer[5] == 1 and layer[2] >= 32):
        continue
    cols = []
    for nt in (0, 1, 2):
        us, gbs, tf = run(layer, nt)
        cols.append(f"nt={nt}: {us:7.1f} us {gbs:7.0f} GB/s {tf:6.0f} TF")
    print(f"{str(layer):44s} " + " | ".join(cols), flush=True)
