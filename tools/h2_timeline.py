#!/usr/bin/env python3
"""Where a workgroup of conv_h2_kernel spends its time (needs an MI355X and a -DH2_TSTAMP library):

    tools/ab_build.sh ts conv_h2_mt2.hip -DH2_TSTAMP=vunet_debug_h2_ts
    VUNET_HIP_LIB=behavior_driven_video_synthesis_amd/build/libvunet_hip_ts.so python tools/h2_timeline.py

Wave 0 of every workgroup records the 100 MHz wall clock at entry (0), after the scale reduction (1), when the first stage is
in LDS (2), after the first chunk (6), after the K loop (3), when the epilogue's stores are issued (4) and retired (5)."""
import ctypes
import os
import sys

os.environ.setdefault("VUNET_ALLOW_TIMING_BUILD", "1")
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from behavior_driven_video_synthesis_amd import ops  # noqa: E402

SHAPES = [
    ("rnb 64ch 128^2 fwd elu", 16, 64, 64, 128, 128, 0, 1),
    ("vgg conv1_2 fwd", 16, 64, 64, 256, 256, 0, 0),
    ("vgg conv2_2 fwd", 16, 128, 128, 128, 128, 0, 0),
    ("vgg conv3_x fwd", 16, 256, 256, 64, 64, 0, 0),
    ("128ch 64^2 dgrad", 16, 128, 128, 64, 64, 1, 0),
]
if len(sys.argv) > 1 and sys.argv[1] == "fill":   # conv1_2 at 1 / 2 / 4 / 16 images: 256 workgroups = ONE per CU, 512 = one round, ...
    SHAPES = [(f"vgg conv1_2 fwd, N = {n_}", n_, 64, 64, 256, 256, 0, 0) for n_ in (1, 2, 4, 16)]
if len(sys.argv) > 1 and sys.argv[1] == "mt1":   # the one-m-tile kernel (conv_h2_mt1.hip built with -DH2_TSTAMP instead)
    SHAPES = [("32ch 256^2 fwd elu", 16, 32, 32, 256, 256, 0, 1), ("32ch 256^2 dgrad", 16, 32, 32, 256, 256, 1, 0)]
lib = ops._lib.lib()
fn = getattr(lib, "vunet_debug_h2_ts")
fn.restype = ctypes.c_int
ops.set_conv_precision("h2")
for name, n, cin, cout, h, w, mode, in_act in SHAPES:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, cin, h, w, generator=g).cuda()
    v = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).cuda()
    wt_f, wt_d, scale, shift, invnorm, wx_f, wx_d = ops.pack_weights(v, None, None, None, None, cin, 0, 1, True)
    mo = cout if mode == 0 else cin
    y = torch.empty(n, mo, h, w, device="cuda")
    d = ops.ConvDesc(N=n, C1=cin if mode == 0 else cout, C2=0, Hs=h, Ws=w, M=mo, m_off=0,
                     Mpad=(wt_f if mode == 0 else wt_d).shape[1], Ho=h, Wo=w, KH=3, KW=3, stride=1, pad=1, mode=mode,
                     in_act=in_act, in_slope=0.0, drop_p=0.0, drop_seed=0, out_act=0, d2s=0)
    wx = wx_f if mode == 0 else wx_d
    amax = ops.absmax_partials(x)

    def launch():
        ops._call("vunet_conv2d_x6", ctypes.byref(d), ops._p(x), None, ops._p(wx), None, None, None, None, ops._p(y),
                  ops._p(amax), None, ops._stream())
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    torch.cuda.synchronize()
    rows = 4 if mo <= 32 else 8   # tile rows: conv_x6.hip's choice for these shapes (MT = 1: NT = 1)
    nb = n * (h // rows) * (w // 32) * ((mo + 63) // 64)
    nb = min(nb, 8192)
    buf = np.zeros((nb, 8), dtype=np.uint64)
    assert fn(buf.ctypes.data_as(ctypes.c_void_p), nb) == 0
    t = buf.astype(np.int64)
    t0 = t[:, 0].min()
    rel = (t - t0) / 100.0   # us
    start, end = rel[:, 0], rel[:, 5]
    print(f"== {name}: {nb} workgroups, launch {e0.elapsed_time(e1) * 1e3:.1f} us, first start -> last end {end.max():.1f} us")
    first = start < 3.0
    for lab, sel in (("first round", first), ("later rounds", ~first)):
        if not sel.any():
            continue
        r = rel[sel]
        seg = {"scale": r[:, 1] - r[:, 0], "first stage": r[:, 2] - r[:, 1], "chunk 0": r[:, 6] - r[:, 2],
               "other chunks": r[:, 3] - r[:, 6], "epilogue issue": r[:, 4] - r[:, 3], "store retire": r[:, 5] - r[:, 4],
               "total": r[:, 5] - r[:, 0]}
        print(f"  {lab} ({int(sel.sum())} workgroups; start {r[:, 0].min():.1f}..{r[:, 0].max():.1f} us)")
        for k_, v_ in seg.items():
            print(f"    {k_:15s} median {np.median(v_):7.2f}  p10 {np.percentile(v_, 10):7.2f}  p90 {np.percentile(v_, 90):7.2f} us")
