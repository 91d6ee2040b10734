"""-m gpu: seeded random sweep of fused-conv configurations (kernel size, stride, padding, ragged channel counts and
map sizes, single / dual source, ELU + dropout prologue, residual) -- forward, data gradient and weight gradient of
the HIP path against the CPU oracle.  Exercises every dispatch branch on shapes nobody hand-picked."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from hip_parity_utils import assert_close, dropout_keep_mask
from synth import seeded_randn, synth_image, synth_param

pytestmark = pytest.mark.gpu


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        k = int(rng.choice([1, 3, 3, 3, 4]))
        stride = int(rng.choice([1, 1, 2]))
        pad = int(rng.choice([0, k // 2, 1])) if k > 1 else 0
        c1 = int(rng.choice([1, 3, 8, 16, 32, 33, 40, 64]))
        dual = bool(rng.integers(0, 3) == 0)
        c2 = int(rng.choice([8, 16, 32])) if dual else 0
        cout = int(rng.choice([1, 3, 16, 32, 35, 64, 96]))
        h = int(rng.choice([4, 7, 8, 16, 20, 32]))
        w = int(rng.choice([4, 9, 16, 32, 64]))
        nb = int(rng.choice([1, 2, 5]))
        if (h + 2 * pad - k) // stride + 1 < 1 or (w + 2 * pad - k) // stride + 1 < 1:
            continue
        act = bool(rng.integers(0, 2))
        drop = float(rng.choice([0.0, 0.0, 0.2])) if act else 0.0
        ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        res = bool(rng.integers(0, 2)) and cout == c1 and (ho, wo) == (h, w)
        out.append((nb, c1, c2, cout, h, w, k, stride, pad, act, drop, res))
    return out


def _ids(c):
    return "-".join(str(int(v) if not isinstance(v, float) else v) for v in c)


@pytest.mark.parametrize("case", _cases(36, 2024), ids=_ids)
def test_fused_conv_random_config_vs_oracle(case):
    _check(case)


# the streaming 1x1 kernel (csrc/conv_1x1.hip) takes channel counts in multiples of 32 on maps too large for split-K:
# one / two / four m-tiles, ragged M, two sources, pixel tiles that straddle rows (W = 40), ELU + dropout, residual
@pytest.mark.parametrize("case", [
    # nb, c1, c2, cout, h,  w, k, stride, pad, act, drop, res
    (2, 32, 0, 32, 32, 32, 1, 1, 0, True, 0.2, True),
    (5, 64, 64, 64, 64, 64, 1, 1, 0, True, 0.0, False),
    (3, 128, 0, 96, 64, 64, 1, 1, 0, False, 0.0, False),
    (5, 32, 32, 128, 48, 40, 1, 1, 0, True, 0.2, False),
    (7, 32, 0, 3, 36, 32, 1, 1, 0, False, 0.0, False),
], ids=_ids)
def test_streaming_1x1_conv_vs_oracle(case):
    from behavior_driven_video_synthesis_amd import ops
    ops.profile_start()
    _check(case)
    kernels = ops.profile_stop(by_kernel=True)
    assert any(k.startswith("conv_1x1_kernel") for k in kernels), kernels


def _check(case):
    from behavior_driven_video_synthesis_amd import ops
    nb, c1, c2, cout, h, w, k, stride, pad, act, drop, res = case
    tag = "fz" + "_".join(map(str, case))
    v = synth_param(tag + ".weight_v", (cout, c1 + c2, k, k), 1)
    g = synth_param(tag + ".weight_g", (cout, 1, 1, 1), 1)
    b = synth_param(tag + ".bias", (cout,), 1)
    gamma = synth_param(tag + ".gamma", (1, cout, 1, 1), 1)
    beta = synth_param(tag + ".beta", (1, cout, 1, 1), 1)
    x1 = synth_image(tag + ".x1", (nb, c1, h, w), 1)
    x2 = synth_image(tag + ".x2", (nb, c2, h, w), 1) if c2 else None
    seed = 4242

    # ---- oracle (plain torch on the CPU)
    ro = [t.clone().requires_grad_(True) for t in (v, g, b, gamma, beta, x1)] + ([x2.clone().requires_grad_(True)] if c2 else [])
    vr, gr, br, gmr, ber, x1r = ro[:6]
    xin = x1r if not c2 else torch.cat([x1r, ro[6]], dim=1)
    a = F.elu(xin) if act else xin
    if drop > 0:
        m = dropout_keep_mask((nb, c1, h, w), drop, seed)
        if c2:
            m = torch.cat([m, dropout_keep_mask((nb, c2, h, w), drop, (seed + ops.SEED2_OFFSET) & 0xFFFFFFFF)], dim=1)
        a = a * m / (1.0 - drop)
    wn = vr * (gr / vr.flatten(1).norm(dim=1).view(-1, 1, 1, 1))
    yr = gmr * F.conv2d(a, wn, br, stride=stride, padding=pad) + ber
    if res:
        yr = yr + x1r
    wgt = seeded_randn(tag + ".wgt", tuple(yr.shape), 1)
    (yr * wgt).sum().backward()

    # ---- HIP
    dev = [t.clone().cuda().requires_grad_(True) for t in (v, g, b, gamma, beta, x1)] + ([x2.clone().cuda().requires_grad_(True)] if c2 else [])
    vd, gd, bd, gmd, bed, x1d = dev[:6]
    cfg = ops.ConvCfg(kind=0, k=k, stride=stride, pad=pad, in_act=ops.ACT_ELU if act else ops.ACT_NONE, drop_p=drop,
                      drop_seed=seed)
    y = ops.fused_conv(x1d, dev[6] if c2 else None, x1d if res else None, vd, gd, bd, gmd, bed, cfg)
    scale = max(1.0, float(yr.abs().max()))
    assert_close(y, yr, rtol=1e-4, atol=1e-4 * scale, name="y")
    (y * wgt.cuda()).sum().backward()
    for name, td, tr in [("dx1", x1d, x1r)] + ([("dx2", dev[6], ro[6])] if c2 else []):
        assert_close(td.grad, tr.grad, rtol=1e-3, atol=1e-4 * max(1.0, float(tr.grad.abs().max())), name=name)
    for name, td, tr in [("dv", vd, vr), ("dg", gd, gr), ("dbias", bd, br), ("dgamma", gmd, gmr), ("dbeta", bed, ber)]:
        assert_close(td.grad, tr.grad, rtol=2e-3, atol=2e-4 * max(1.0, float(tr.grad.abs().max())), name=name)


# the VALU kernels for layers with 3 channels on one side (csrc/conv_thin.hip): 3x3 with M = 3 (forward and, through
# the backward of a 3 -> C layer, the data gradient C -> 3) and the 1x1 input layer 3 -> C; maps >= 64K pixels
@pytest.mark.parametrize("case", [
    # nb, c1, c2, cout, h,   w, k, stride, pad, act, drop, res
    (2, 32, 0, 3, 192, 176, 3, 1, 1, False, 0.0, False),     # out_conv: thin_m forward (ragged tiles: 176 = 5.5 x 32)
    (4, 3, 0, 64, 128, 128, 3, 1, 1, False, 0.0, False),     # conv1_1: its data gradient 64 -> 3 is thin_m mode 1
    (3, 3, 0, 32, 160, 144, 1, 1, 0, False, 0.0, False),     # nin 3 -> 32: thin_k 1x1
    (2, 3, 0, 40, 200, 168, 3, 1, 1, False, 0.0, False),     # 3 -> 40, 3x3: thin_k forward, ragged M and tiles
    (2, 32, 0, 3, 256, 128, 3, 1, 1, False, 0.0, False),     # out_conv at tile-aligned size: thin_k data gradient (3 -> 32,
                                                             # mirrored taps) and the FMA weight-gradient kernel
    (1, 64, 0, 4, 264, 256, 3, 1, 1, True, 0.0, False),      # 64 -> 4 with an ELU prologue: three items per thread
], ids=_ids)
def test_three_channel_side_kernels_vs_oracle(case):
    from behavior_driven_video_synthesis_amd import ops
    ops.profile_start()
    _check(case)
    kernels = ops.profile_stop(by_kernel=True)
    if not case[9]:   # (a prologue keeps the forward / data gradient off the VALU kernels)
        assert any(k.startswith("conv_thin_") for k in kernels), kernels
    if case[3] <= 4 and case[4] % 8 == 0 and case[5] % 32 == 0:
        assert any(k.startswith("conv_wgrad_thin_kernel") for k in kernels), kernels
    if case[3] == 3 and case[6] == 3:
        assert any(k.startswith("conv_thin_kv_kernel<3") for k in kernels), kernels     # the 3 -> C data gradient
