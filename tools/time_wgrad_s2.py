#!/usr/bin/env python3
"""A/B of the stride-2 weight gradient on the step's Downsample layers (bs 16): the LDS-staged conv_wgrad_s2_kernel against the
direct kernel (tuning knob wgrad_rowsplit = 3).  HIP events, alternating in one process."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from behavior_driven_video_synthesis_amd import ops
    for n, c1, cout, hs in ((16, 32, 64, 256), (16, 64, 128, 128)):
        ho = hs // 2
        x = torch.randn(n, c1, hs, hs, device="cuda")
        dy = torch.randn(n, cout, ho, ho, device="cuda")
        amx, amd = ops.absmax_partials(x, None), ops.absmax_partials(dy)
        res = {}
        for rep in range(3):
            for knob in (0, 3):
                ops.set_tuning("wgrad_rowsplit", knob)
                wd = ops.WgradDesc(N=n, C1=c1, C2=0, Hs=hs, Ws=hs, Cout=cout, Ho=ho, Wo=ho, KH=3, KW=3, stride=2, pad=1,
                                   in_act=ops.ACT_NONE, in_slope=0.0, drop_p=0.0, drop_seed=0, nsplit=1, flags=2)
                wd.nsplit = ops._lib.lib().vunet_conv2d_wgrad_nsplit(ctypes.byref(wd))
                cp, ktot = ops._r32(cout), 9 * c1
                slabs = torch.empty(wd.nsplit * cp * (ktot + 1), device="cuda")
                dshift = slabs[wd.nsplit * cp * ktot:]

                def run():
                    ops._call("vunet_conv2d_wgrad", ctypes.byref(wd), ops._p(x), None, ops._p(dy), ops._p(slabs), ops._p(dshift),
                              ops._p(amx), ops._p(amd), ops._stream())
                for _ in range(3):
                    run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    run()
                e1.record()
                torch.cuda.synchronize()
                res.setdefault(knob, []).append(round(e0.elapsed_time(e1) / 20 * 1e3, 1))
                res[f"nsplit_{knob}"] = wd.nsplit
        flop = 2.0 * n * ho * ho * cout * 9 * c1
        print(json.dumps({"layer": f"{c1}->{cout} @ {hs}^2 -> {ho}^2, bs {n}", "staged_us": res[0], "direct_us": res[3],
                          "nsplit_staged": res["nsplit_0"], "nsplit_direct": res["nsplit_3"],
                          "staged_TFLOPs": round(flop / min(res[0]) / 1e6, 1), "direct_TFLOPs": round(flop / min(res[3]) / 1e6, 1)}))


if __name__ == "__main__":
    main()
