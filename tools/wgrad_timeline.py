#!/usr/bin/env python3
"""Where a group of conv_wgrad_h2_kernel spends a step (needs an MI355X and a -DWG_TSTAMP library):

    tools/ab_build.sh wgts conv_wgrad_h2.hip -DWG_TSTAMP=vunet_debug_wg_ts
    VUNET_HIP_LIB=behavior_driven_video_synthesis_amd/build/libvunet_hip_wgts.so python tools/wgrad_timeline.py

Wave 0 of each group stamps the 100 MHz wall clock in its third step: 0 top, 1 f(x) loads returned, 2 converted + written to
LDS, 3 past the barrier, 4 dy loads returned, 5 MFMA block done, 6 past the closing barrier.  Medians over workgroups, us."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from behavior_driven_video_synthesis_amd import ops  # noqa: E402

lib = ops._lib.lib()
fn = getattr(lib, "vunet_debug_wg_ts")
fn.restype = ctypes.c_int
ops.set_conv_precision("h2")
for name, n, c1, cout, h, w in [("128ch 128^2", 16, 128, 128, 128, 128), ("64ch 256^2", 16, 64, 64, 256, 256),
                                ("32ch 256^2", 16, 32, 32, 256, 256)]:
    x = torch.randn(n, c1, h, w, device="cuda")
    dy = torch.randn(n, cout, h, w, device="cuda")
    wd = ops.WgradDesc(N=n, C1=c1, C2=0, Hs=h, Ws=w, Cout=cout, Ho=h, Wo=w, KH=3, KW=3, stride=1, pad=1, in_act=1, in_slope=0.0,
                       drop_p=0.0, drop_seed=0, nsplit=1, flags=2)
    ns = lib.vunet_conv2d_wgrad_nsplit(ctypes.byref(wd))
    wd.nsplit = ns
    ktot = 9 * c1
    slabs = torch.empty(ns * ops._r32(cout) * (ktot + 1), device="cuda")
    dshift = slabs[ns * ops._r32(cout) * ktot:]
    ax, ad = ops.absmax_partials(x), ops.absmax_partials(dy)
    for _ in range(int(os.environ.get("WG_TL_LAUNCHES", "3"))):   # (30: the sustained, clock-limited regime of back-to-back launches)
        ops._call("vunet_conv2d_wgrad", ctypes.byref(wd), ops._p(x), None, ops._p(dy), ops._p(slabs), ops._p(dshift), ops._p(ax),
                  ops._p(ad), ops._stream())
    torch.cuda.synchronize()
    nb = min(4096, ns * (c1 // 32) * ((cout + 63) // 64 if cout % 64 == 0 else (cout + 31) // 32))
    buf = np.zeros(nb * 16, dtype=np.uint64)
    assert fn(buf.ctypes.data_as(ctypes.c_void_p), nb) == 0
    t = buf.reshape(nb, 2, 8).astype(np.int64)
    ent, loop_end, ext = t[:, 0, 7], t[:, 1, 7], t[:, 1, 6]
    print(f"{name}: nsplit {ns}, {nb} workgroups; per group, median us since the step's top (100 MHz clock)")
    print(f"  workgroup: entry -> tile loop over {np.median((loop_end - ent) / 100.0):7.1f} us, -> exit {np.median((ext - ent) / 100.0):7.1f} us; "
          f"first entry -> last exit {(ext.max() - ent.min()) / 100.0:7.1f} us; entries spread over {(ent.max() - ent.min()) / 100.0:6.1f} us")
    for g in range(2):
        d = (t[:, g, 1:7] - t[:, g, 0:1]) / 100.0
        ok = (t[:, g, 0] > 0) & (t[:, g, 6] > 0)
        if ok.sum() == 0:
            continue
        med = np.median(d[ok], axis=0)
        print(f"  group {g}: loads {med[0]:6.2f}  converted {med[1]:6.2f}  barrier {med[2]:6.2f}  dy {med[3]:6.2f}  mfma {med[4]:6.2f}  "
              f"barrier {med[5]:6.2f}   (offset of group 1's top against group 0's: "
              f"{np.median((t[ok, 1, 0] - t[ok, 0, 0]) / 100.0) if g == 1 else 0:6.2f})")
