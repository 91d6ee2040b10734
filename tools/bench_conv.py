#!/usr/bin/env python3
"""Micro-benchmark of one fused-conv shape (forward, data gradient, weight gradient) through the C-ABI.

    python tools/bench_conv.py N C1 C2 H W M k stride [--act 1] [--drop 0.05] [--res] [--iters 20]
Environment knobs of the library (VUNET_NO_TILED, VUNET_TILED_FORCE_NT) apply.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from behavior_driven_video_synthesis_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("dims", type=int, nargs=8)
ap.add_argument("--act", type=int, default=0)
ap.add_argument("--drop", type=float, default=0.0)
ap.add_argument("--res", action="store_true")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--sparse", action="store_true", help="ReLU-like inputs (about half zeros), as the VGG layers see")
a = ap.parse_args()
N, C1, C2, H, W, M, k, s = a.dims
dev = "cuda:0"
x1 = torch.randn(N, C1, H, W, device=dev)
if a.sparse:
    x1 = torch.relu(x1)
x1.requires_grad_(True)
x2 = torch.randn(N, C2, H, W, device=dev, requires_grad=True) if C2 else None
v = torch.randn(M, C1 + C2, k, k, device=dev, requires_grad=True)
g = torch.rand(M, 1, 1, 1, device=dev, requires_grad=True)
b = torch.randn(M, device=dev, requires_grad=True)
gamma = torch.ones(1, M, 1, 1, device=dev, requires_grad=True)
beta = torch.zeros(1, M, 1, 1, device=dev, requires_grad=True)
pad = k // 2
res = x1 if (a.res and M == C1 and s == 1) else None


def step():
    cfg = ops.ConvCfg(kind=0, k=k, stride=s, pad=pad, in_act=a.act, drop_p=a.drop, drop_seed=1234)
    y = ops.fused_conv(x1, x2, res, v, g, b, gamma, beta, cfg)
    y.backward(torch.ones_like(y))


for _ in range(3):
    step()
torch.cuda.synchronize()
ops.profile_start()
for _ in range(a.iters):
    step()
torch.cuda.synchronize()
fam = ops.profile_stop()
for name, f in sorted(fam.items()):
    print(f"{name:18s} {1e3 * f['ms'] / f['n']:9.1f} us/launch  {f['flop'] / (f['ms'] * 1e-3) / 1e12:7.1f} TF/s  ({f['n'] // a.iters} launches/iter)")
