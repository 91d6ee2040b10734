#!/usr/bin/env python3
"""Render-loop benchmark (BASELINE config 5): T-frame pose sequence -> GPU stickman raster -> VunetAlter.transfer.

    python tools/bench_render.py [--frames 50] [--size 256] [--chunk 25] [--iters 5]
Prints one JSON line per mode: fp32, bf16 operands, bf16 + one appearance encoding per sequence; for the bf16
kernel also its achieved HBM rate (algorithmic bytes: fp32 inputs + outputs + residual of each launch).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from behavior_driven_video_synthesis_amd import ops  # noqa: E402
from behavior_driven_video_synthesis_amd.experiments.shape_and_pose_net import DEFAULT_CONFIG  # noqa: E402
from behavior_driven_video_synthesis_amd.models.vunets import VunetAlter  # noqa: E402
from behavior_driven_video_synthesis_amd.render import render_sequence  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=50)
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--chunk", type=int, default=50)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--layers", action="store_true", help="per-layer kernel times of the bf16 modes on stderr")
ap.add_argument("--no-ref", action="store_true", help="skip the fp32 reference render (no PSNR): profiling runs")
ap.add_argument("--only", default="", help="run just this mode (the PMC passes of tools/profile.sh)")
a = ap.parse_args()

torch.manual_seed(0)
kw = dict(DEFAULT_CONFIG["architecture"])
kw.update(DEFAULT_CONFIG["data"])
kw["spatial_size"] = a.size
net = VunetAlter(n_channels_x=3, dropout_prob=0.05, **kw).cuda().eval()
app = (torch.rand(1, 3, a.size, a.size, device="cuda") * 2 - 1)
kps = torch.rand(a.frames, 17, 2, device="cuda") * (a.size - 20) + 10
with torch.no_grad():
    eps = [torch.randn_like(m) for m in net.appearance_code(app)]


def blk_bytes(key):
    """Algorithmic bytes of one conv_blk launch: bf16 sources + output (+ residual); the fp32 output layer 4 B/value."""
    _, n, c1, c2, hs, ws, m, k, s, has_res, nchw, _ = key
    ho, wo = (hs - 1) // s + 1, (ws - 1) // s + 1
    return 2.0 * n * hs * ws * (c1 + c2) + (4.0 if nchw else 2.0) * n * ho * wo * m + (2.0 * n * ho * wo * m if has_res else 0.0)


def run(**kwargs):
    for _ in range(2):
        out, _ = render_sequence(net, app, kps, chunk=a.chunk, as_uint8=False, eps=eps, **kwargs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        out, _ = render_sequence(net, app, kps, chunk=a.chunk, as_uint8=False, eps=eps, **kwargs)
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) / a.iters


ref, t32 = (None, 0.0) if a.no_ref else run()
peak = 0.0 if ref is None else 2 * float(ref.abs().max())
MODES = (("fp32", {}), ("bf16 operands, fp32 NCHW activations", {"dtype": "bf16", "layout": "nchw"}),
         ("bf16 operands, fp32 NCHW activations+shared_appearance", {"dtype": "bf16", "layout": "nchw", "share_appearance": True}),
         ("bf16 blocked", {"dtype": "bf16"}),
         ("bf16 blocked+shared_appearance", {"dtype": "bf16", "share_appearance": True}),
         ("fp32+shared_appearance", {"share_appearance": True}))
if a.only:
    MODES = tuple(m for m in MODES if m[0] == a.only)
for name, kwargs in MODES:
    out, t = run(**kwargs)
    mse = 0.0 if ref is None else float(((out - ref) ** 2).mean())
    rec = {"mode": name, "frames": a.frames, "size": a.size, "ms_per_sequence": round(1e3 * t, 2),
           "frames_per_s": round(a.frames / t, 1),
           "psnr_vs_fp32_db": None if mse == 0 else round(10 * torch.log10(torch.tensor(peak * peak / mse)).item(), 1)}
    if "bf16" in name:
        ops.profile_start()
        render_sequence(net, app, kps, chunk=a.chunk, as_uint8=False, eps=eps, **kwargs)
        fam = ops.profile_stop(detail=True)
        tot = sum(v["ms"] for v in fam.values())
        per_kernel = {}
        for key, v in fam.items():
            if key[0] == "conv_bf16_fwd":
                _, n, c1, c2, hs, ws, m, k, s, act, kern = key
                per = 4.0 * n * hs * ws * (c1 + c2 + m + (m if act else 0))   # RNB layers (ELU prologue) add the residual
            elif key[0] == "conv_blk_fwd":
                per, kern = blk_bytes(key), key[-1]
            else:
                continue
            e = per_kernel.setdefault(kern, {"ms": 0.0, "bytes": 0.0, "flop": 0.0, "n": 0})
            e["ms"] += v["ms"]
            e["bytes"] += per * v["n"]
            e["flop"] += v["flop"]
            e["n"] += v["n"]
        for kern, e in per_kernel.items():
            rec[kern] = {"ms_per_sequence": round(e["ms"], 2), "launches": e["n"], "share_of_conv_time": round(e["ms"] / tot, 3),
                         "algorithmic_GBps": round(e["bytes"] / (e["ms"] * 1e-3) / 1e9, 1),
                         "frac_of_hbm_peak": round(e["bytes"] / (e["ms"] * 1e-3) / 8e12, 3),
                         "TFLOPs": round(e["flop"] / (e["ms"] * 1e-3) / 1e12, 1)}
        if a.layers:
            for key, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])[:24]:
                print("#", key, round(v["ms"], 3), "ms", v["n"], "launches", file=sys.stderr)
    print(json.dumps(rec))
