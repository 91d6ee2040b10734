#!/usr/bin/env python3
"""A/B of the blocked render path's skip blocks: one fused launch (vunet_conv2d_blk_rnb) vs the 1x1 + 3x3 pair, on the render
loop's shapes (50 frames).  HIP events, same process, alternating."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))


def main():
    from behavior_driven_video_synthesis_amd.lib.modules import VunetRNB
    from behavior_driven_video_synthesis_amd.render_blk import BlockedTransfer, blk_empty

    class _Net:
        pass
    for c, hw in ((32, 256), (64, 128)):
        blk = VunetRNB(channels=c, a_channels=c, residual=True, dropout_prob=0.0).cuda().eval()
        eng = BlockedTransfer.__new__(BlockedTransfer)
        eng.vunet, eng._packs, eng.fuse_rnb = _Net(), {}, True
        x = blk_empty(50, c, hw, hw, "cuda")
        a = blk_empty(50, c, hw, hw, "cuda")
        x.view(torch.int16).random_(-9000, 9000)
        a.view(torch.int16).random_(-9000, 9000)
        res = {}
        for rep in range(3):
            for fuse in (True, False):
                eng.fuse_rnb = fuse
                for _ in range(3):
                    eng._rnb(blk, x, a)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    eng._rnb(blk, x, a)
                e1.record()
                torch.cuda.synchronize()
                res.setdefault(fuse, []).append(e0.elapsed_time(e1) / 20 * 1e3)
        elems = 50 * c * hw * hw
        print(json.dumps({"channels": c, "map": hw, "frames": 50, "fused_us": [round(v, 1) for v in res[True]],
                          "two_launches_us": [round(v, 1) for v in res[False]],
                          "fused_GBps": round(2 * elems * 4 / min(res[True]) / 1e3, 1),
                          "two_launches_GBps": round(2 * elems * 6 / min(res[False]) / 1e3, 1)}))


if __name__ == "__main__":
    main()
