#!/usr/bin/env python3
"""The perceptual loss's gradient and feature taps against the CPU oracle evaluated in float64: HIP h2 (split fp16), HIP f32
(fp32-input MFMA), and the float32 CPU oracle itself.    python tools/vgg_grad_cmp.py [size] [width_div] [batch]
The gradient of an L1 loss on ReLU / max-pool features is discontinuous in the features, so every float32 evaluation
scatters around the float64 one at the 1e-3 level; this prints who scatters how much."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden")]
from synth import synth_image  # noqa: E402
from behavior_driven_video_synthesis_amd import ops  # noqa: E402
from behavior_driven_video_synthesis_amd.lib.losses import vgg_loss  # noqa: E402
from behavior_driven_video_synthesis_amd.models.imagenet_pretrained import PerceptualVGG, vgg19  # noqa: E402
from oracle import vunet_oracle as O  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
wdiv = int(sys.argv[2]) if len(sys.argv) > 2 else 2
bs = int(sys.argv[3]) if len(sys.argv) > 3 else 2
seed = 78 if (size, wdiv) == (256, 1) else 77
tag = "vl256" if (size, wdiv) == (256, 1) else "vl"
torch.set_num_threads(min(torch.get_num_threads(), 32))
weights = [1.0, 0.5, 1.5, 1.0, 2.0, 1.0]
pv = PerceptualVGG(vgg19(seed=seed, width_div=wdiv), weights).cuda()
vsd = O.make_synthetic_vgg19(seed=seed, width_div=wdiv)
target = synth_image(tag + ".t", (bs, 3, size, size), 5)
pred0 = synth_image(tag + ".p", (bs, 3, size, size), 6)


def ref(dtype):
    p = pred0.clone().to(dtype).requires_grad_(True)
    sd = {k: v.to(dtype) for k, v in vsd.items()}
    feats = {k: v.detach().double() for k, v in O.perceptual_vgg(sd, p, last=31).items()}
    ld = O.vgg_loss(sd, weights, target.to(dtype), p)
    torch.stack([v.sum() for v in ld.values()]).sum().backward()
    return p.grad.double(), feats


def hip(mode):
    ops.set_conv_precision(mode)
    p = pred0.cuda().requires_grad_(True)
    with torch.no_grad():
        feats = {k: v.double().cpu() for k, v in pv(p.detach()).items()}
    ld = vgg_loss(pv, target.cuda(), p)
    torch.stack([v.sum() for v in ld.values()]).sum().backward()
    return p.grad.double().cpu(), feats


def rel(a, b):
    return float((a - b).norm() / b.norm()), float((a - b).abs().max() / b.abs().max())


(g32, f32c), (g64, f64c) = ref(torch.float32), ref(torch.float64)
(gh2, fh2), (gf32, ff32) = hip("h2"), hip("f32")
print(f"VGG19 width / {wdiv}, {size}x{size}, batch {bs}: (relative L2, max / max) against the float64 oracle")
print("gradient  cpu fp32", rel(g32, g64))
print("gradient  hip h2  ", rel(gh2, g64))
print("gradient  hip f32 ", rel(gf32, g64))
print("gradient  hip h2 vs hip f32", rel(gh2, gf32), " hip h2 vs cpu fp32", rel(gh2, g32))
for k in f64c:
    print(f"feature {k:8s} cpu fp32 {rel(f32c[k], f64c[k])[0]:.2e}  hip h2 {rel(fh2[k], f64c[k])[0]:.2e}  hip f32 {rel(ff32[k], f64c[k])[0]:.2e}")
