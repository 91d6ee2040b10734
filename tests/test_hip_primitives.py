"""-m gpu: HIP primitives (through the C-ABI library) vs golden vectors from the reference and vs the oracle.

Tolerance: fp32 MFMA is a k-ordered fmaf chain (exact fp32), the CPU reference sums in a different
order -> atol/rtol 1e-4 on outputs, 1e-3 on weight gradients (long reductions)  (SURVEY 8c).
"""
import ctypes

import pytest
import torch

from conftest import load_golden
from hip_parity_utils import assert_close, dropout_keep_mask
from synth import seeded_randn, synth_image, synth_state_dict

pytestmark = pytest.mark.gpu


def _mods():
    from behavior_driven_video_synthesis_amd.lib import modules as M
    return M


def _build(case):
    M = _mods()
    return {
        "nc_k3s1": lambda: M.NormConv2d(6, 10, 3, 1, 1),
        "nc_k1": lambda: M.NormConv2d(3, 8, 1),
        "nc_k3valid": lambda: M.NormConv2d(3, 16, 3),
        "down": lambda: M.Downsample(8, 16),
        "up": lambda: M.Upsample(8, 4),
        "rnb_plain": lambda: M.VunetRNB(8),
        "rnb_res": lambda: M.VunetRNB(8, a_channels=8, residual=True),
        "rnb_res2": lambda: M.VunetRNB(8, a_channels=16, residual=True),
        "s2d": lambda: M.SpaceToDepth(2),
        "d2s": lambda: M.DepthToSpace(2),
        "l2nc": lambda: M.L2NormConv2d(6, 8, 3, 1, 1, bias=False),
        "lnc": lambda: M.LayerNormConv2d(6, 8, 3, 1, 1),
    }[case]()


@pytest.mark.parametrize("case", ["k3", "k3s2", "k1"])
def test_l2norm_data_dependent_init_vs_golden(case):
    """L2NormConv2d with ``init_fn() -> True`` (lib/modules.py:95-99) on the GPU against the values the reference module
    produced: gamma, beta, the output of the initialising forward and of the next one (VERDICT r2 #8a)."""
    from behavior_driven_video_synthesis_amd.lib import modules as Mm
    meta, arr = load_golden("g1c_l2norm_init")
    seed, info = meta["seed"], meta["cases"][case]
    cin, cout, k, stride, pad = info["args"]
    flag = {"on": True}
    mod = Mm.L2NormConv2d(cin, cout, k, stride, pad, bias=False, init=lambda: flag["on"])
    assert {k_: list(v.shape) for k_, v in mod.state_dict().items()} == info["shapes"]
    mod.load_state_dict(synth_state_dict(info["shapes"], seed))
    mod = mod.cuda().train()
    y = mod(synth_image(f"l2i.{case}.x", tuple(info["input"]), seed).cuda())
    assert_close(mod.gamma, arr[f"{case}.gamma"], rtol=1e-4, atol=1e-5, name="gamma")
    assert_close(mod.beta, arr[f"{case}.beta"], rtol=1e-4, atol=1e-5, name="beta")
    assert_close(y, arr[f"{case}.y_init"], name="y_init")
    flag["on"] = False
    y2 = mod(synth_image(f"l2i.{case}.x2", tuple(info["input"]), seed).cuda())
    assert_close(y2, arr[f"{case}.y_after"], name="y_after")


CASES = ["nc_k3s1", "nc_k1", "nc_k3valid", "down", "up", "rnb_plain", "rnb_res", "rnb_res2", "s2d", "d2s", "l2nc",
         "lnc"]


@pytest.mark.parametrize("case", CASES)
def test_primitive_vs_golden(case):
    meta, arr = load_golden("g1_primitives")
    seed, info = meta["seed"], meta["cases"][case]
    mod = _build(case)
    assert {k: list(v.shape) for k, v in mod.state_dict().items()} == info["shapes"]
    mod.load_state_dict(synth_state_dict(info["shapes"], seed))
    mod = mod.cuda().train()
    names = [case + ".x", case + ".a"]
    ins = [synth_image(names[i], tuple(s), seed).cuda().requires_grad_(True) for i, s in enumerate(info["inputs"])]
    y = mod(*ins)
    assert_close(y, arr[case + ".y"], name=case + ".y")
    (y * seeded_randn(case + ".wgt", tuple(y.shape), seed).cuda()).sum().backward()
    for i, t in enumerate(ins):
        assert_close(t.grad, arr[f"{case}.gin{i}"], name=f"{case}.gin{i}")
    for k, p in mod.named_parameters():
        assert_close(p.grad, arr[f"{case}.gp.{k}"], rtol=1e-3, atol=1e-4, name=f"{case}.gp.{k}")


@pytest.mark.parametrize("shape", [
    # (N, Cin, Cout, H, W, k, stride, pad)
    (2, 32, 32, 32, 32, 3, 1, 1),     # full 32-channel tile
    (1, 64, 128, 16, 16, 3, 1, 1),    # multiple m-blocks
    (3, 5, 70, 9, 11, 3, 1, 1),       # ragged channels, ragged map
    (2, 16, 24, 17, 13, 3, 2, 1),     # stride 2, odd sizes
    (2, 3, 8, 10, 10, 4, 2, 1),       # PatchGAN-style 4x4 stride 2
    (2, 8, 4, 6, 6, 4, 1, 1),         # 4x4 stride 1
    (4, 128, 128, 4, 4, 3, 1, 1),     # bottleneck 4x4 map: tiles span images
    (2, 7, 9, 2, 2, 3, 1, 1),         # 2x2 map (Market config latents)
    (1, 16, 3, 40, 40, 3, 1, 1),      # out_conv-like: 3 output channels
    (2, 32, 32, 8, 32, 3, 1, 1),      # LDS halo-tile wgrad path, one m-tile per workgroup
    (1, 64, 128, 12, 64, 3, 1, 1),    # LDS halo-tile wgrad path, two m-tiles, several column tiles
    (3, 32, 64, 4, 96, 3, 1, 1),      # tiled wgrad, more tiles than splits
    (2, 32, 3, 8, 64, 3, 1, 1),       # tiled wgrad with a partial output-channel tile (out_conv)
    (2, 32, 32, 8, 64, 1, 1, 0),      # 1x1 (nin) through the tiled wgrad kernel
    (2, 64, 128, 4, 32, 1, 1, 0),     # 1x1, two m-tiles per workgroup
    (2, 3, 32, 8, 64, 1, 1, 0),       # first nin: 3 input channels, masked channel block
    (8, 32, 64, 64, 128, 3, 2, 1),    # stride-2 Downsample through the LDS-tiled kernel (128 workgroups)
    (16, 16, 32, 32, 128, 3, 2, 1),   # stride-2, one m-tile
])
def test_normconv_vs_oracle(shape):
    """Fused NormConv2d fwd + dgrad + wgrad on shapes the golden file does not hold (edge cases)."""
    from oracle import vunet_oracle as O
    M = _mods()
    n, cin, cout, h, w, k, s, p = shape
    torch.manual_seed(0)
    mod = M.NormConv2d(cin, cout, k, s, p)
    sd = synth_state_dict({kk: list(v.shape) for kk, v in mod.state_dict().items()}, 5)
    mod.load_state_dict(sd)
    x = synth_image("x", (n, cin, h, w), 5)
    sdr = {"m." + kk: v.clone().requires_grad_(True) for kk, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    yr = O.norm_conv(sdr, "m", xr, s, p)
    wgt = seeded_randn("w", tuple(yr.shape), 5)
    (yr * wgt).sum().backward()
    mod = mod.cuda()
    xg = x.cuda().requires_grad_(True)
    y = mod(xg)
    assert_close(y, yr, name="y")
    (y * wgt.cuda()).sum().backward()
    assert_close(xg.grad, xr.grad, name="dx")
    for kk, pp in mod.named_parameters():
        assert_close(pp.grad, sdr["m." + kk].grad, rtol=1e-3, atol=2e-4, name=kk)


def test_rnb_dropout_matches_oracle_with_same_mask():
    """Dropout is a stateless hash of (element index, seed): the oracle is fed the same keep-mask."""
    from oracle import vunet_oracle as O
    from behavior_driven_video_synthesis_amd import ops
    M = _mods()
    c, p = 16, 0.3
    mod = M.VunetRNB(c, a_channels=c, residual=True, dropout_prob=p)
    sd = synth_state_dict({k: list(v.shape) for k, v in mod.state_dict().items()}, 7)
    mod.load_state_dict(sd)
    mod = mod.cuda().train()
    x, a = synth_image("x", (2, c, 12, 12), 7), synth_image("a", (2, c, 12, 12), 7)
    ops.set_dropout_seed(123)
    seed = ops.next_dropout_seed()
    ops.set_dropout_seed(123)  # the module will draw the same seed
    xg, ag = x.cuda().requires_grad_(True), a.cuda().requires_grad_(True)
    y = mod(xg, ag)
    m1 = dropout_keep_mask((2, c, 12, 12), p, seed)
    m2 = dropout_keep_mask((2, c, 12, 12), p, (seed + ops.SEED2_OFFSET) & 0xFFFFFFFF)
    assert_close(ops.dropout_keep_mask((2, c, 12, 12), p, seed, "cuda"), m1, 0, 0, "mask")
    mask = torch.cat([m1, m2], dim=1)
    assert 0.5 < float(mask.mean()) < 0.9
    sdr = {"m." + k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr, ar = x.clone().requires_grad_(True), a.clone().requires_grad_(True)
    yr = O.rnb(sdr, "m", xr, ar, drop_mask=mask, drop_p=p)
    assert_close(y, yr, name="y")
    wgt = seeded_randn("w", tuple(yr.shape), 7)
    (yr * wgt).sum().backward()
    (y * wgt.cuda()).sum().backward()
    assert_close(xg.grad, xr.grad, name="dx")
    assert_close(ag.grad, ar.grad, name="da")
    for k, pp in mod.named_parameters():
        assert_close(pp.grad, sdr["m." + k].grad, rtol=1e-3, atol=2e-4, name=k)


def test_rnb_dual_source_tiled_wgrad_vs_oracle():
    """Residual block at 32 channels / 32-wide maps: dual-source conv through the tiled wgrad kernel, dropout on."""
    from oracle import vunet_oracle as O
    from behavior_driven_video_synthesis_amd import ops
    M = _mods()
    c, p = 32, 0.1
    mod = M.VunetRNB(c, a_channels=c, residual=True, dropout_prob=p)
    sd = synth_state_dict({k: list(v.shape) for k, v in mod.state_dict().items()}, 9)
    mod.load_state_dict(sd)
    mod = mod.cuda().train()
    x, a = synth_image("x", (2, c, 8, 64), 9), synth_image("a", (2, c, 8, 64), 9)
    ops.set_dropout_seed(77)
    seed = ops.next_dropout_seed()
    ops.set_dropout_seed(77)
    xg, ag = x.cuda().requires_grad_(True), a.cuda().requires_grad_(True)
    y = mod(xg, ag)
    mask = torch.cat([dropout_keep_mask((2, c, 8, 64), p, seed),
                      dropout_keep_mask((2, c, 8, 64), p, (seed + ops.SEED2_OFFSET) & 0xFFFFFFFF)], dim=1)
    sdr = {"m." + k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr, ar = x.clone().requires_grad_(True), a.clone().requires_grad_(True)
    yr = O.rnb(sdr, "m", xr, ar, drop_mask=mask, drop_p=p)
    assert_close(y, yr, name="y")
    wgt = seeded_randn("w", tuple(yr.shape), 9)
    (yr * wgt).sum().backward()
    (y * wgt.cuda()).sum().backward()
    assert_close(xg.grad, xr.grad, name="dx")
    assert_close(ag.grad, ar.grad, name="da")
    for k, pp in mod.named_parameters():
        assert_close(pp.grad, sdr["m." + k].grad, rtol=1e-3, atol=2e-4, name=k)


@pytest.mark.parametrize("nt", [1, 2, 4, 16])
@pytest.mark.parametrize("cin,cout,dual", [(32, 32, False), (16, 64, True), (64, 128, False), (8, 3, False)])
def test_lds_tiled_conv_vs_oracle(nt, cin, cout, dual, monkeypatch):
    """LDS-tiled 3x3 kernel (forward with ELU+dropout prologue, data gradient with mirrored taps), every tile
    height, one and two m-tiles, single and dual source; forced on small tensors through ops.set_tuning("tiled_force_nt")."""
    from oracle import vunet_oracle as O
    from behavior_driven_video_synthesis_amd import ops
    M = _mods()
    ops.set_tuning("tiled_force_nt", nt)
    n, h, w, p = 2, 16, (64 if nt != 16 else 16), 0.1
    if nt == 16:  # 16-wide maps: two 16-pixel row segments per MFMA pixel tile
        ops.set_tuning("tiled_force_nt", 1)
    if dual:
        mod = M.VunetRNB(cout, a_channels=cin, residual=True, dropout_prob=p)
    else:
        mod = M.NormConv2d(cin, cout, 3, 1, 1)
    sd = synth_state_dict({k: list(v.shape) for k, v in mod.state_dict().items()}, 13)
    mod.load_state_dict(sd)
    mod = mod.cuda().train()
    sdr = {"m." + k: v.clone().requires_grad_(True) for k, v in sd.items()}
    if dual:
        x, a = synth_image("x", (n, cout, h, w), 13), synth_image("a", (n, cin, h, w), 13)
        ops.set_dropout_seed(5)
        seed = ops.next_dropout_seed()
        ops.set_dropout_seed(5)
        xg, ag = x.cuda().requires_grad_(True), a.cuda().requires_grad_(True)
        y = mod(xg, ag)
        mask = torch.cat([dropout_keep_mask((n, cout, h, w), p, seed),
                          dropout_keep_mask((n, cout, h, w), p, (seed + ops.SEED2_OFFSET) & 0xFFFFFFFF)], dim=1)
        xr, ar = x.clone().requires_grad_(True), a.clone().requires_grad_(True)
        yr = O.rnb(sdr, "m", xr, ar, drop_mask=mask, drop_p=p)
        pairs = [(xg, xr, "dx"), (ag, ar, "da")]
    else:
        x = synth_image("x", (n, cin, h, w), 13)
        xg, xr = x.cuda().requires_grad_(True), x.clone().requires_grad_(True)
        y, yr = mod(xg), O.norm_conv(sdr, "m", xr, 1, 1)
        pairs = [(xg, xr, "dx")]
    assert_close(y, yr, name="y")
    wgt = seeded_randn("w", tuple(yr.shape), 13)
    (yr * wgt).sum().backward()
    (y * wgt.cuda()).sum().backward()
    for g, r, name in pairs:
        assert_close(g.grad, r.grad, name=name)
    for k, pp in mod.named_parameters():
        assert_close(pp.grad, sdr["m." + k].grad, rtol=1e-3, atol=2e-4, name=k)


def test_cpu_tensors_are_refused():
    M = _mods()
    mod = M.NormConv2d(4, 4, 3, 1, 1)
    with pytest.raises(RuntimeError):
        mod(torch.zeros(1, 4, 8, 8))


def test_frozen_weight_pack_follows_the_weights():
    """Frozen layers (the VGG19 stack) pack their weights once; the pack must never outlive or mismatch them: a new
    layer that lands on a freed layer's address, and an in-place weight update, both have to be seen."""
    import torch.nn.functional as F
    M = _mods()
    x = seeded_randn("frozen.x", (2, 8, 16, 16), 1).cuda()

    def check(layer):
        with torch.no_grad():
            y = layer(x)
        ref = F.conv2d(x.cpu(), layer.weight.detach().cpu(), layer.bias.detach().cpu(), padding=1)
        assert_close(y, ref, rtol=1e-4, atol=1e-5, name="frozen conv")

    for i in range(4):   # allocator reuse: each new layer is likely to sit where the previous one was
        layer = M.Conv2d(8, 8, 3, 1, 1).cuda().requires_grad_(False)
        with torch.no_grad():
            layer.weight.copy_(seeded_randn(f"frozen.w{i}", (8, 8, 3, 3), 1))
        check(layer)
        check(layer)     # second call: served from the pack
        with torch.no_grad():
            layer.weight.mul_(-0.5)   # in-place update bumps the version counter
            layer.bias.add_(1.0)
        check(layer)
        del layer


@pytest.mark.parametrize("case", [
    # N, C1, C2,  H,  W,   M, in_act, res, d2s, out_act
    (2, 16, 0, 8, 32, 16, 0, False, False, 0),
    (2, 32, 16, 16, 32, 64, 0, False, False, 0),      # two sources, two m-tiles
    (1, 48, 0, 12, 64, 96, 0, False, False, 0),       # M = 96: the second 64-wide block is half empty
    (3, 16, 16, 32, 32, 32, 1, True, False, 0),       # ELU prologue + residual (the RNB layer)
    (2, 32, 0, 16, 32, 64, 1, False, True, 0),        # sub-pixel up-conv: depth-to-space store
    (2, 16, 0, 64, 32, 24, 0, False, False, 3),       # tall map, sigmoid epilogue, ragged M
], ids=lambda c: "-".join(str(v) for v in c))
def test_bf16_inference_conv_vs_rounded_operand_reference(case):
    """vunet_conv2d_bf16: the result equals an fp64 convolution of the bf16-rounded operands (layout / indexing
    check at accumulation-order tolerance); ELU runs before the rounding, as in the kernel."""
    import torch.nn.functional as F
    from behavior_driven_video_synthesis_amd import ops
    n, c1, c2, h, w, m, in_act, use_res, d2s, out_act = case
    x1 = seeded_randn("bf.x1", (n, c1, h, w), 5).cuda()
    x2 = seeded_randn("bf.x2", (n, c2, h, w), 5).cuda() if c2 else None
    v = (seeded_randn("bf.v", (m, c1 + c2, 3, 3), 5) * 0.1).cuda()
    b = seeded_randn("bf.b", (m,), 5).cuda()
    res = seeded_randn("bf.res", (n, m, h, w), 5).cuda() if use_res else None
    cfg = ops.ConvCfg(kind=1, k=3, stride=1, pad=1, in_act=ops.ACT_ELU if in_act else ops.ACT_NONE,
                      out_act=out_act, d2s=d2s)
    ops.profile_start()
    with torch.no_grad(), ops.inference_precision("bf16"):
        y = ops.fused_conv(x1, x2, res, v, None, b, None, None, cfg)
    assert "conv_bf16_fwd" in ops.profile_stop()

    def r16(t):
        return t.to(torch.bfloat16).double()
    x = x1 if x2 is None else torch.cat([x1, x2], 1)
    x = x.cpu()
    if in_act:
        x = F.elu(x)
    ref = F.conv2d(r16(x), r16(v.cpu()), b.cpu().double(), padding=1)
    if out_act == ops.ACT_SIGMOID:
        ref = torch.sigmoid(ref)
    if d2s:
        ref = F.pixel_shuffle(ref.reshape(n, 4, m // 4, h, w).transpose(1, 2).reshape(n, m, h, w), 2)
    if use_res:
        ref = ref + res.cpu().double()
    # fp32 accumulation order; with ELU a few inputs may round to the neighbouring bf16 value (fast-exp ulps)
    tol = 2e-3 if in_act else 2e-5
    assert_close(y, ref.float(), rtol=tol, atol=tol * float(ref.abs().max()), name="bf16 conv")


@pytest.mark.parametrize("case", [
    # N, Cin, Cout, H,  W, k, stride  -- frozen conv + ReLU; dx through vunet_conv2d_dgrad_relu where the tiled kernel applies
    (16, 64, 64, 64, 64, 3, 1),      # 16-row tiles
    (16, 32, 128, 32, 32, 3, 1),     # 8-row tiles
    (4, 64, 96, 16, 32, 3, 1),       # 4-row tiles, ragged M
    (16, 64, 64, 16, 16, 3, 1),      # 16-wide maps (two row segments per MFMA tile)
    (2, 16, 24, 12, 20, 3, 1),       # geometry the tiled kernel does not take: two-pass route
    (3, 32, 32, 32, 32, 1, 1),       # 1x1: two-pass route
], ids=lambda c: "-".join(map(str, c)))
def test_relu_data_gradient_of_frozen_layers_vs_autograd(case):
    import torch.nn.functional as F
    from behavior_driven_video_synthesis_amd import ops
    n, cin, cout, h, w, k, stride = case
    x = seeded_randn("rd.x", (n, cin, h, w), 2)
    v = seeded_randn("rd.v", (cout, cin, k, k), 2) * (1.0 / (cin * k * k) ** 0.5)
    b = seeded_randn("rd.b", (cout,), 2) * 0.1
    wgt = seeded_randn("rd.w", (n, cout, (h + 2 * (k // 2) - k) // stride + 1, (w + 2 * (k // 2) - k) // stride + 1), 2)
    xr = x.clone().requires_grad_(True)
    (F.relu(F.conv2d(xr, v, b, stride=stride, padding=k // 2)) * wgt).sum().backward()
    xd = x.clone().cuda().requires_grad_(True)
    cfg = ops.ConvCfg(kind=1, k=k, stride=stride, pad=k // 2, out_act=ops.ACT_RELU)
    y = ops.fused_conv(xd, None, None, v.cuda(), None, b.cuda(), None, None, cfg)   # weights do not require grad: frozen
    (y * wgt.cuda()).sum().backward()
    assert_close(xd.grad, xr.grad, rtol=1e-3, atol=1e-4 * float(xr.grad.abs().max()), name="dx")


def test_weightnorm_backward_ignores_stale_output_buffers():
    """Non-accumulate mode must not read its output buffers: torch.empty blocks recycled by the caching allocator can
    hold NaN / inf, and ``0 * NaN`` is NaN.  Pre-fill every gradient output with NaN and compare with zeros."""
    import ctypes
    from behavior_driven_video_synthesis_amd import ops
    torch.manual_seed(3)
    cout, cin, k, ns = 24, 10, 3, 3
    K = cin * k * k
    v = torch.randn(cout, cin, k, k, device="cuda")
    g, bias = torch.rand(cout, 1, 1, 1, device="cuda") + 0.5, torch.randn(cout, device="cuda")
    gamma = torch.rand(1, cout, 1, 1, device="cuda") + 0.5
    slabs = torch.randn(ns * 32 * K + ns * 32, device="cuda")
    dshift = slabs[ns * 32 * K:]
    invnorm = 1.0 / v.flatten(1).norm(dim=1)

    def run(fill):
        outs = [torch.full_like(t, fill) for t in (v, g, bias, gamma, gamma)]
        work = torch.empty(cout * (K + 1), device="cuda")
        wn = ops.WnDesc(cout, cin, 0, k, k, 0, 0)
        ops._call("vunet_weightnorm_bwd", ctypes.byref(wn), ops._p(slabs), ops._p(dshift), ns, ops._p(v), ops._p(g),
                  ops._p(bias), ops._p(gamma), ops._p(invnorm), *[ops._p(o) for o in outs], ops._p(work), 0, ops._stream())
        return outs
    a, b = run(float("nan")), run(0.0)
    for x, y in zip(a, b):
        assert torch.isfinite(x).all() and torch.equal(x, y)


def test_l2_data_dependent_init_keeps_parameters_in_the_optimizer_bucket():
    """lib/modules.py:95-99 under FusedAdam: gamma / beta are views of the flat bucket; the init must write through them
    (rebinding .data detached them -- Adam kept updating a slot the module no longer read)."""
    from behavior_driven_video_synthesis_amd.lib.modules import L2NormConv2d
    from behavior_driven_video_synthesis_amd.optim import FusedAdam
    from oracle import vunet_oracle as O
    flag = {"on": True}
    conv = L2NormConv2d(6, 10, 3, padding=1, bias=False, init=lambda: flag["on"]).cuda().train()
    opt = FusedAdam(list(conv.parameters()), lr=1e-2)
    x = torch.randn(4, 6, 12, 12, device="cuda")
    y = conv(x)
    b = opt.buckets[0]
    for p in (conv.gamma, conv.beta):
        off = p.data_ptr() - b.flat.data_ptr()
        assert 0 <= off < 4 * b.numel, "parameter left the flat bucket"
    # the initialised layer normalises its output over (N, H, W): mean 0, variance 1 per channel
    assert float(y.mean(dim=(0, 2, 3)).abs().max()) < 1e-4 and float((y.var(dim=(0, 2, 3)) - 1).abs().max()) < 1e-3
    y0 = O.l2norm_conv({".weight": conv.weight.detach().cpu(), ".gamma": torch.ones(1, 10, 1, 1),
                        ".beta": torch.zeros(1, 10, 1, 1)}, "", x.cpu(), 1, 1)
    assert_close(conv.gamma.detach().cpu(), 1.0 / torch.sqrt(y0.var(dim=[0, 2, 3], keepdim=True) + 1e-10), rtol=1e-4,
                 atol=1e-5, name="gamma init")
    flag["on"] = False
    g0 = conv.gamma.detach().clone()
    conv(x).square().mean().backward()
    opt.step()
    assert not torch.equal(conv.gamma.detach(), g0)          # Adam's update is what the module reads
    assert conv.gamma.data_ptr() - b.flat.data_ptr() < 4 * b.numel


def test_bilinear_upsample_branch_vs_golden():
    """Upsample(subpixel=False) and a VunetAlter with subpixel_upsampling False on the HIP path (lib/modules.py:172-182,
    models/vunets.py:325-329) against the reference's outputs."""
    from behavior_driven_video_synthesis_amd.lib.modules import Upsample
    from behavior_driven_video_synthesis_amd.models.vunets import VunetAlter
    meta, arr = load_golden("g1b_upsample_bilinear")
    seed = meta["seed"]
    up = Upsample(8, 6, subpixel=False)
    assert {k: list(v.shape) for k, v in up.state_dict().items()} == meta["shapes"]
    up.load_state_dict(synth_state_dict(meta["shapes"], seed))
    up = up.cuda().train()
    x = synth_image("upb.x", (2, 8, 7, 10), seed).cuda().requires_grad_(True)
    y = up(x)
    assert_close(y, arr["y"], name="y")
    (y * seeded_randn("upb.w", tuple(y.shape), seed).cuda()).sum().backward()
    assert_close(x.grad, arr["gx"], rtol=1e-3, atol=1e-5, name="gx")
    for k, p_ in up.named_parameters():
        assert_close(p_.grad, arr["gp." + k], rtol=1e-3, atol=1e-4, name=k)
    net = VunetAlter(n_channels_x=3, **meta["cfg"])
    assert {k: list(v.shape) for k, v in net.state_dict().items()} == meta["model_shapes"]
    net.load_state_dict(synth_state_dict(meta["model_shapes"], seed))
    net = net.cuda().train()
    xi, c = synth_image("upb.mx", (2, 3, 32, 32), seed).cuda(), synth_image("upb.mc", (2, 3, 32, 32), seed).cuda()
    eps = [seeded_randn(f"upb.eps{i}", tuple(s), seed).cuda() for i, s in enumerate(meta["eps_shapes"])]
    img, _, _, _ = net(xi, c, eps)
    assert_close(img, arr["img"], name="img")
    (img * seeded_randn("upb.mw", tuple(img.shape), seed).cuda()).sum().backward()
    params = dict(net.named_parameters())
    for k, s in meta["grad_sums"].items():
        if s is not None:
            got = float(params[k].grad.double().abs().sum())
            assert abs(got - s[1]) <= 1e-3 * s[1] + 1e-4, (k, got, s[1])


def test_l1_tap_and_pool_backward_in_one_pass():
    """vunet_l1_pool_bwd (the VGG taps that feed a loss term AND a max-pool) == vunet_maxpool2_bwd followed by
    vunet_l1_mean_bwd with its `add` input, bit for bit, with and without the ReLU mask; maxima published."""
    import ctypes
    from behavior_driven_video_synthesis_amd import ops
    g = torch.Generator().manual_seed(21)
    n, c, h, w = 2, 5, 12, 10
    pred = torch.relu(torch.randn(n, c, h, w, generator=g)).cuda()          # a ReLU output: zeros and ties included
    target = torch.relu(torch.randn(n, c, h, w, generator=g)).cuda()
    dyp = torch.randn(n, c, h // 2, w // 2, generator=g).cuda()
    gout = torch.tensor([0.7], device="cuda")
    y = torch.empty(n, c, h // 2, w // 2, device="cuda")
    ops._call("vunet_maxpool2_fwd", ops._p(pred), ops._p(y), n * c, h, w, ops._stream())
    scale = 1.5 / pred.numel()
    for mask in (0, 1):
        dx = torch.empty_like(pred)
        ops._call("vunet_maxpool2_bwd_relu" if mask else "vunet_maxpool2_bwd", ops._p(pred), ops._p(y), ops._p(dyp), ops._p(dx),
                  n * c, h, w, ops._stream())
        want = torch.empty_like(pred)
        ops._call("vunet_l1_mean_bwd_amax", ops._p(target), ops._p(pred), ops._p(dx), ops._p(want), scale, ops._p(gout),
                  pred.numel(), None, mask, ops._stream())
        got = torch.full_like(pred, float("nan"))
        amax = torch.zeros(1024, device="cuda")
        ops._call("vunet_l1_pool_bwd", ops._p(target), ops._p(pred), ops._p(dyp), ops._p(got), scale, ops._p(gout), n * c, h, w,
                  ops._p(amax), mask, ops._stream())
        assert torch.equal(got, want)
        assert float(amax.max()) == float(want.abs().max())
        if mask:
            assert float(got[pred <= 0].abs().max()) == 0.0
    # and against autograd on the CPU (the unmasked form is exactly d/dpred of  w * mean|t - p| + <maxpool(p), dy>)
    p_ = pred.cpu().clone().requires_grad_(True)
    loss = 1.5 * (target.cpu() - p_).abs().mean() * 0.7 + (torch.nn.functional.max_pool2d(p_, 2) * dyp.cpu()).sum()
    loss.backward()
    got = torch.empty_like(pred)
    ops._call("vunet_l1_pool_bwd", ops._p(target), ops._p(pred), ops._p(dyp), ops._p(got), scale, ops._p(gout), n * c, h, w,
              None, 0, ops._stream())
    nz = (pred != target).cpu()       # (sign(0) is 0 here and in ATen alike; ties of the max go to the first element in both)
    assert torch.allclose(got.cpu()[nz], p_.grad[nz], rtol=1e-6, atol=1e-7)


def test_l1_tap_and_pool_forward_in_one_pass():
    """vunet_l1_pool_fwd == vunet_l1_mean_fwd + vunet_maxpool2_fwd (the pooled tensor bit for bit, the loss to the rounding of
    a different summation order)."""
    from behavior_driven_video_synthesis_amd import ops
    g = torch.Generator().manual_seed(22)
    n, c, h, w = 3, 7, 20, 14
    pred = torch.randn(n, c, h, w, generator=g).cuda()
    target = torch.randn(n, c, h, w, generator=g).cuda()
    partial = torch.empty(1024, device="cuda")
    want_l = torch.zeros(1, device="cuda")
    ops._call("vunet_l1_mean_fwd", ops._p(target), ops._p(pred), ops._p(partial), ops._p(want_l), 1.5, pred.numel(), ops._stream())
    want_y = torch.empty(n, c, h // 2, w // 2, device="cuda")
    ops._call("vunet_maxpool2_fwd", ops._p(pred), ops._p(want_y), n * c, h, w, ops._stream())
    got_l = torch.zeros(1, device="cuda")
    got_y = torch.full_like(want_y, float("nan"))
    ops._call("vunet_l1_pool_fwd", ops._p(target), ops._p(pred), ops._p(partial), ops._p(got_l), ops._p(got_y), 1.5, n * c, h, w,
              ops._stream())
    assert torch.equal(got_y, want_y)
    assert abs(float(got_l) - float(want_l)) <= 2e-6 * abs(float(want_l))
    assert abs(float(got_l) - 1.5 * float((target - pred).abs().double().mean())) <= 2e-6 * abs(float(want_l))


@pytest.mark.parametrize("mode,act,drop,with_res,c,cout", [
    (0, "none", 0.0, True, 32, 32),      # forward, no output activation, residual
    (0, "relu", 0.0, False, 64, 128),    # forward + ReLU, two m-tiles ... four
    (1, "none", 0.0, True, 32, 32),      # data gradient, no activation derivative
    (1, "elu", 0.0, True, 64, 64),       # data gradient * ELU'(aux) + res
    (1, "elu", 0.1, True, 32, 64),       # ... with the forward pass's dropout mask
    (0, "sigmoid", 0.0, False, 32, 32),  # the general form stays reachable
])
@pytest.mark.gpu
def test_1x1_kernel_16_byte_epilogue_forms(mode, act, drop, with_res, c, cout):
    """The streaming 1x1 kernel's 16-byte epilogue (conv_common.h: store_tile_side4, FORM 0..3) against fp64:
    forward y = act(W x + shift) + res, data gradient dx = W^T dy * ELU'(aux) * keep / (1 - p) + res."""
    import ctypes
    import torch.nn.functional as F
    from behavior_driven_video_synthesis_amd import ops
    n, h, w, seed = 2, 8, 64, 123
    g = torch.Generator().manual_seed(c + cout + mode)
    v = (torch.randn(cout, c, 1, 1, generator=g) * 0.2).cuda()
    bias = torch.randn(cout, generator=g).cuda()
    wt_f, wt_d, scale, shift, invnorm, wx_f, wx_d = ops.pack_weights(v, None, bias, None, None, c, 0, 1, True)
    cin_k, mo = (c, cout) if mode == 0 else (cout, c)          # channels read / written by this launch
    x = torch.randn(n, cin_k, h, w, generator=g).cuda()
    res = torch.randn(n, mo, h, w, generator=g).cuda() if with_res else None
    aux = torch.randn(n, mo, h, w, generator=g).cuda() if (mode == 1 and act == "elu") else None
    out_act = {"none": 0, "relu": ops.ACT_RELU, "sigmoid": ops.ACT_SIGMOID, "elu": 0}[act] if mode == 0 else 0
    d = ops.ConvDesc(N=n, C1=cin_k, C2=0, Hs=h, Ws=w, M=mo, m_off=0, Mpad=(wt_f if mode == 0 else wt_d).shape[1], Ho=h, Wo=w,
                     KH=1, KW=1, stride=1, pad=0, mode=mode, in_act=0, in_slope=0.0, drop_p=0.0, drop_seed=0, out_act=out_act,
                     d2s=0, aux_act=ops.ACT_ELU if aux is not None else 0, aux_slope=0.0, aux_drop_p=drop, aux_drop_seed=seed)
    y = torch.full((n, mo, h, w), float("nan"), device="cuda")
    ops._call("vunet_conv2d_gather", ctypes.byref(d), ops._p(x), None, ops._p(wt_f if mode == 0 else wt_d),
              ops._p(shift) if mode == 0 else None, ops._p(res), ops._p(aux), ops._p(y), ops._stream())
    wd = v.double().cpu() * scale.double().cpu().view(-1, 1, 1, 1)
    if mode == 0:
        ref = F.conv2d(x.double().cpu(), wd) + shift.double().cpu().view(1, -1, 1, 1)
        ref = {"none": ref, "relu": ref.clamp_min(0), "sigmoid": torch.sigmoid(ref)}[act]
    else:
        ref = F.conv_transpose2d(x.double().cpu(), wd)
        if aux is not None:
            a = aux.double().cpu()
            ref = ref * torch.where(a > 0, torch.ones_like(a), a.exp())
            if drop > 0:
                ref = ref * dropout_keep_mask((n, mo, h, w), drop, seed).double() * float(torch.tensor(1.0 / (1.0 - drop),
                                                                                                      dtype=torch.float32))
    if res is not None:
        ref = ref + res.double().cpu()
    assert float((y.double().cpu() - ref).abs().max()) <= 2e-5 * max(float(ref.abs().max()), 1.0)


@pytest.mark.parametrize("shape", [(2, 3, 8, 8), (1, 5, 6, 24), (2, 4, 4, 12), (3, 32, 64, 64), (1, 2, 2, 2)])
def test_space_to_depth_kernel_forms(shape):
    """vunet_space_to_depth in both kernel forms (16-byte vectors for W % 8 == 0, per element otherwise) against the
    index formula of lib/modules.py:11-21: y[n, (2i + j) C + c, h, w] = x[n, c, 2h + i, 2w + j] -- a permutation, bit-exact;
    and depth-to-space inverts it."""
    from behavior_driven_video_synthesis_amd import ops
    n, c, h, w = shape
    x = torch.randn(n, c, h, w, device="cuda")
    y = ops.SpaceToDepth.apply(x)
    ref = x.view(n, c, h // 2, 2, w // 2, 2).permute(0, 3, 5, 1, 2, 4).reshape(n, 4 * c, h // 2, w // 2)
    assert torch.equal(y, ref)
    assert torch.equal(ops.DepthToSpace.apply(y), x)


@pytest.mark.parametrize("use_kl,gamma", [(True, "dev"), (True, 1.0), (False, "dev")])
def test_total_loss_and_gamma_controller_vs_the_reference_expressions(use_kl, gamma):
    """ops.TotalLoss / ops.gamma_update_ (one launch each) against the reference's expression chains
    (experiments/shape_and_pose_net.py:391-405 and :82-85): values to 1 ulp of fp32, gradients exactly."""
    from behavior_driven_video_synthesis_amd import ops
    g = torch.Generator().manual_seed(5)
    vals = (torch.rand(6, generator=g) * 40).tolist()
    terms = [torch.tensor([v], device="cuda", requires_grad=True) for v in vals]
    kl = torch.tensor(3.25, device="cuda", requires_grad=True)
    gam = torch.tensor(0.37, device="cuda") if gamma == "dev" else gamma
    loss, ll = ops.TotalLoss.apply(kl, gam, 0.7, use_kl, *terms)
    t2 = [t.detach().clone().requires_grad_(True) for t in terms]
    kl2 = kl.detach().clone().requires_grad_(True)
    ll_ref = 0.7 * torch.sum(torch.stack(t2, dim=0))
    loss_ref = ll_ref + gam * kl2 if use_kl else ll_ref
    assert loss.shape == loss_ref.shape == () and ll.shape == ()
    assert abs(float(ll) - float(ll_ref)) <= 2e-7 * abs(float(ll_ref))
    assert abs(float(loss) - float(loss_ref)) <= 2e-7 * abs(float(loss_ref))
    # both outputs carry gradient (the adversarial term's adaptive weight differentiates likelihood_loss on its own)
    (2.0 * loss + 0.5 * ll).backward()
    (2.0 * loss_ref + 0.5 * ll_ref).backward()
    for a, b in zip(terms, t2):
        assert a.grad.shape == b.grad.shape and torch.equal(a.grad, b.grad)
    if use_kl:
        assert torch.equal(kl.grad, kl2.grad)
    else:
        assert kl.grad is None and kl2.grad is None
    for g0, imax, klv in ((0.5, 2.0, 3.0), (0.001, 9.0, 1.0), (0.0, 1.0, 1.0)):
        gd = torch.tensor(g0, device="cuda")
        im, kv = torch.tensor(imax, device="cuda"), torch.tensor([klv], device="cuda")
        want = torch.clamp(gd - 0.01 * (im - kv.reshape(())), min=0.0)
        ops.gamma_update_(gd, im, kv, 0.01)
        assert torch.equal(gd, want)


def test_unit_sample_draws_standard_normal_noise_and_passes_the_gradient_through():
    """ops.UnitSample (vunet_unit_sample): z - mu is N(0, 1) noise -- moments, tail mass and lag-1 correlation of 2^20 draws
    within a few standard errors -- reproducible from the seed sequence, fresh per call and per device step; dz/dmu = 1."""
    from behavior_driven_video_synthesis_amd import ops
    n = 1 << 20
    mu = seeded_randn("us.mu", (4, 64, 64, 64), 3).cuda().requires_grad_(True)
    ops.set_dropout_seed(1234)
    z1 = ops.UnitSample.apply(mu)
    z2 = ops.UnitSample.apply(mu)
    ops.set_dropout_seed(1234)
    z1b = ops.UnitSample.apply(mu)
    assert torch.equal(z1, z1b) and not torch.equal(z1, z2)
    e = (z1 - mu).detach().double().flatten()
    assert abs(float(e.mean())) < 5 / n ** 0.5
    assert abs(float(e.var()) - 1.0) < 5 * (2.0 / n) ** 0.5
    assert abs(float((e ** 3).mean())) < 5 * (15.0 / n) ** 0.5
    assert abs(float((e ** 4).mean()) - 3.0) < 5 * (96.0 / n) ** 0.5
    assert abs(float((e.abs() > 2.0).double().mean()) - 0.0455003) < 5 * (0.0455 / n) ** 0.5
    assert abs(float((e[1:] * e[:-1]).mean())) < 5 / n ** 0.5
    assert abs(float((e * (z2 - mu).detach().double().flatten()).mean())) < 5 / n ** 0.5
    g = seeded_randn("us.g", tuple(mu.shape), 4).cuda()
    z1.backward(g)
    assert torch.equal(mu.grad, g)
    # with a log-std (reparametrize): z = eps * exp(logstd) + mu, eps kept for the backward, gradients as vunet_reparam_bwd
    ls = (0.3 * seeded_randn("us.ls", tuple(mu.shape), 5)).cuda().requires_grad_(True)
    mu2 = mu.detach().clone().requires_grad_(True)
    ops.set_dropout_seed(77)
    zr = ops.Reparam.apply(mu2, ls, None)
    er = ((zr - mu2) / ls.exp()).detach().double().flatten()
    assert abs(float(er.mean())) < 5 / n ** 0.5 and abs(float(er.var()) - 1.0) < 5 * (2.0 / n) ** 0.5
    zr.backward(g)
    assert torch.equal(mu2.grad, g)
    assert_close(ls.grad, g * (zr - mu2).detach(), 1e-5, 1e-6, "dlogstd")
    # the device step counter moves the noise without changing a launch argument
    ctr = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.set_dropout_step(ctr)
    try:
        ops.set_dropout_seed(99)
        a = ops.UnitSample.apply(mu.detach())
        ctr.fill_(1)
        ops.set_dropout_seed(99)
        b = ops.UnitSample.apply(mu.detach())
        ctr.fill_(0)
        ops.set_dropout_seed(99)
        a2 = ops.UnitSample.apply(mu.detach())
    finally:
        ops.set_dropout_step(None)
    assert torch.equal(a, a2) and not torch.equal(a, b)


@pytest.mark.parametrize("n", [2, 3, 4])
def test_fan_out_sums_its_readers_gradients_in_one_launch(n):
    """ops.fan_out / vunet_sum_amax: the gradient of a tensor with n readers equals the autograd engine's sum bit for bit
    (the engine accumulates later-created readers first), and the sum carries its |.| maxima as a tag."""
    from behavior_driven_video_synthesis_amd import ops
    x = seeded_randn("fo.x", (3, 16, 8, 8), 11).cuda().requires_grad_(True)
    ws = [seeded_randn(f"fo.w{i}", (3, 16, 8, 8), 12 + i).cuda() * (10.0 ** (i - 1)) for i in range(n)]
    hs = ops.fan_out(x * 1.0, n)
    assert len(hs) == n and all(h.data_ptr() == hs[0].data_ptr() for h in hs)
    seen = {}
    hs[0].grad_fn.next_functions   # (the aliases share ONE backward node)
    y = sum((h * w).sum() for h, w in zip(hs, ws))
    mid = hs[0]._base if hs[0]._base is not None else hs[0]
    mid.register_hook(lambda g: seen.setdefault("g", g))
    y.backward()
    x2 = x.detach().clone().requires_grad_(True)
    h2 = x2 * 1.0
    sum((h2 * w).sum() for w in ws).backward()
    assert torch.equal(x.grad, x2.grad)
    g = seen["g"]
    tag = ops._tagged_amax(g)
    assert tag is not None and float(tag.max()) == float(g.abs().max())
    # the raw entry point: argument checks
    lib = ops._lib.lib()
    out = torch.empty_like(ws[0])
    ptrs = (ctypes.c_void_p * 1)(ws[0].data_ptr())
    assert lib.vunet_sum_amax(ptrs, 1, ops._p(out), None, out.numel(), ops._stream()) < 0
