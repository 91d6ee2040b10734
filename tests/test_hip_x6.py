"""-m gpu: the fp32-accurate split convolution kernels against an fp64 convolution, every case on BOTH schemes:

  h2  two scaled fp16 terms / three partial products, csrc/conv_h2_kernel.h + conv_wgrad_h2.hip  (the default)
  x6  three bf16 terms / six partial products,        csrc/conv_x6_kernel.h + conv_wgrad_x6.hip

The claim under test: splitting both fp32 operands into 16-bit terms whose partial products are exact in fp32 and
accumulating them in fp32 on the matrix cores gives fp32-LEVEL accuracy (what is dropped is <= 2^-22 / 2^-24 of the
result; what remains is accumulation rounding).  Measured on MI355X (tools/x6_accuracy.py ->
profiles/r02_split_accuracy.txt): rms error of h2 2.4e-8 .. 9.4e-8, of x6 3.6e-8 .. 2.3e-7 of the output scale for
K = 144 .. 4608 (the fp32-MFMA kernel's fmaf chain: 2.7e-8 .. 6.8e-8), more than 100x below the 1e-4 parity tolerance.
So every case is compared (i) with an fp64 reference at a bound of 3e-6 of the output scale -- 30x tighter than the
tolerance the fp32-MFMA kernels are held to -- and (ii) with the fp32-MFMA kernel's own error on the same problem (never
worse than 5x that error plus the rounding floor).  The fp16 scheme's data-dependent pieces (|x| maxima, their
producer-side tags, the weight image's scale exponent, inputs spanning 2^-23 in magnitude, 16-wide maps) have their own
cases below.
"""
import ctypes
import os

import pytest
import torch
import torch.nn.functional as F

from hip_parity_utils import assert_close, dropout_keep_mask

pytestmark = pytest.mark.gpu


def _ops():
    from behavior_driven_video_synthesis_amd import ops
    return ops


@pytest.fixture(autouse=True, params=["x6", "h2"])
def scheme(request):
    """Every test of this module runs on both split schemes: three bf16 terms / six products (conv_x6_kernel.h) and two
    scaled fp16 terms / three products (conv_h2_kernel.h).  Same bounds for both."""
    ops = _ops()
    before = ops.conv_precision()
    ops.set_conv_precision(request.param)
    yield request.param
    ops.set_conv_precision(before)


def _amax(ops, x1, x2=None):
    """The |x| partial maxima the fp16 scheme's kernels scale by (NULL for the bf16 scheme)."""
    return ops._p(ops.absmax_partials(x1, x2)) if ops.conv_precision() == "h2" else None


def _ref_forward(x1, x2, v, scale, shift, in_act, drop, seed, out_act, res, d2s):
    """fp64 restatement of one fused layer: prologue on each source, conv, shift, activation, d2s, residual."""
    ops = _ops()
    xs = []
    for i, x in enumerate((x1, x2)):
        if x is None:
            continue
        t = x.double().cpu()
        if in_act == ops.ACT_ELU:
            t = F.elu(t)
        if drop > 0:
            s = seed if i == 0 else (seed + ops.SEED2_OFFSET) & 0xFFFFFFFF
            t = t * dropout_keep_mask(tuple(x.shape), drop, s).double() * float(torch.tensor(1.0 / (1.0 - drop),
                                                                                             dtype=torch.float32))
        xs.append(t)
    xin = torch.cat(xs, dim=1)
    w = (v.double().cpu() * scale.double().cpu().view(-1, 1, 1, 1))
    y = F.conv2d(xin, w, None, padding=1) + shift.double().cpu().view(1, -1, 1, 1)
    if out_act == ops.ACT_RELU:
        y = torch.relu(y)
    elif out_act == ops.ACT_SIGMOID:
        y = torch.sigmoid(y)
    if d2s:
        from oracle import vunet_oracle as O
        y = O.depth_to_space(y)
    if res is not None:
        y = y + res.double().cpu()
    return y


def _run_forward(ops, x1, x2, v, g, bias, gamma, beta, in_act, drop, seed, out_act, res, d2s, use_x6, force_nt=None):
    n, c1, h, w = x1.shape
    c2 = 0 if x2 is None else x2.shape[1]
    cout = v.shape[0]
    wt_f, wt_d, scale, shift, invnorm, wx_f, wx_d = ops.pack_weights(v, g, bias, gamma, beta, c1, c2, 0, True)
    assert wx_f is not None
    y = torch.empty((n, cout // 4, 2 * h, 2 * w) if d2s else (n, cout, h, w), device="cuda")
    d = ops.ConvDesc(N=n, C1=c1, C2=c2, Hs=h, Ws=w, M=cout, m_off=0, Mpad=wt_f.shape[1], Ho=h, Wo=w, KH=3, KW=3,
                     stride=1, pad=1, mode=0, in_act=in_act, in_slope=0.0, drop_p=drop, drop_seed=seed,
                     out_act=out_act, d2s=int(d2s))
    if force_nt is not None:
        ops.set_tuning("split_force_nt", force_nt)
    try:
        if use_x6 and w % 32 and ops.conv_precision() != "h2":   # 16-pixel column tiles exist in the fp16 scheme only
            rc = ops._lib.lib().vunet_conv2d_x6(ctypes.byref(d), ops._p(x1), ops._p(x2), ops._p(wx_f), ops._p(shift),
                                                ops._p(res), None, None, ops._p(y), None, None, ops._stream())
            assert rc == -3
            pytest.skip("16-wide maps: fp16 scheme only (the bf16 kernel refuses them)")
        if use_x6:
            assert w % 32 != 0 or ops._lib.lib().vunet_conv2d_x6_supported(ctypes.byref(d), 0) == 1
            ops._call("vunet_conv2d_x6", ctypes.byref(d), ops._p(x1), ops._p(x2), ops._p(wx_f), ops._p(shift),
                      ops._p(res), None, None, ops._p(y), _amax(ops, x1, x2), None, ops._stream())
        else:
            ops._call("vunet_conv2d_gather", ctypes.byref(d), ops._p(x1), ops._p(x2), ops._p(wt_f), ops._p(shift),
                      ops._p(res), None, ops._p(y), ops._stream())
    finally:
        ops.set_tuning("split_force_nt", 0)
    torch.cuda.synchronize()
    return y, scale, shift, (wt_d, wx_d)


def _params(cout, cin, seed):
    g_ = torch.Generator().manual_seed(seed)
    v = (torch.randn(cout, cin, 3, 3, generator=g_) * 0.2).cuda()
    g = (torch.rand(cout, 1, 1, 1, generator=g_) + 0.5).cuda()
    bias = (torch.randn(cout, generator=g_) * 0.1).cuda()
    gamma = (1.0 + 0.3 * torch.randn(1, cout, 1, 1, generator=g_)).cuda()
    beta = (0.2 * torch.randn(1, cout, 1, 1, generator=g_)).cuda()
    return v, g, bias, gamma, beta


# (n, c1, c2, cout, h, w, in_act, drop, out_act, with_res, d2s, forced NT)
FWD_CASES = [
    (2, 32, 0, 32, 16, 32, 0, 0.0, 0, False, False, 4),    # MT 1, 16-row tile
    (2, 32, 0, 32, 8, 64, 1, 0.0, 0, True, False, 2),      # MT 1, 8 rows, ELU prologue + residual
    (1, 16, 0, 32, 4, 32, 1, 0.1, 0, True, False, 1),      # one chunk, dropout, 4-row tile
    (2, 32, 32, 32, 16, 32, 1, 0.05, 0, True, False, 4),   # two sources (the RNB skip read), ELU + dropout
    (2, 64, 0, 64, 8, 32, 0, 0.0, 2, False, False, 2),     # MT 2, ReLU epilogue (VGG19 layer)
    (1, 64, 64, 64, 8, 64, 1, 0.05, 0, True, False, 2),    # MT 2, two sources
    (1, 128, 0, 256, 8, 32, 0, 0.0, 0, False, True, 2),    # sub-pixel up-conv: depth-to-space store, 4 m-blocks
    (1, 48, 16, 96, 12, 32, 1, 0.0, 3, False, False, 1),   # ragged: 3 + 1 chunks, M = 96 (last m-block half empty), sigmoid
    (1, 256, 0, 128, 8, 32, 0, 0.0, 0, False, False, 2),   # long K loop (16 chunks)
    (2, 64, 0, 64, 16, 16, 0, 0.0, 2, False, False, 1),    # 16-wide map (VGG19 conv5): two rows per column tile, fp16 scheme only
    (1, 32, 32, 32, 8, 16, 1, 0.05, 0, True, False, 1),    # 16-wide, two sources, ELU + dropout, residual, MT 1
    (3, 16, 0, 96, 24, 48, 1, 0.0, 0, True, False, 1),     # 48 wide: three 16-pixel column tiles per row pair
]


@pytest.mark.parametrize("case", FWD_CASES)
def test_x6_forward_vs_fp64_and_vs_fp32_mfma(case):
    ops = _ops()
    n, c1, c2, cout, h, w, in_act, drop, out_act, with_res, d2s, nt = case
    g_ = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x1 = torch.randn(n, c1, h, w, generator=g_).cuda()
    x2 = torch.randn(n, c2, h, w, generator=g_).cuda() if c2 else None
    res = None
    if with_res:
        res = torch.randn((n, cout // 4, 2 * h, 2 * w) if d2s else (n, cout, h, w), generator=g_).cuda()
    v, g, bias, gamma, beta = _params(cout, c1 + c2, 5)
    seed = 0xC0FFEE
    y6, scale, shift, _ = _run_forward(ops, x1, x2, v, g, bias, gamma, beta, in_act, drop, seed, out_act, res, d2s,
                                       True, nt)
    y32, _, _, _ = _run_forward(ops, x1, x2, v, g, bias, gamma, beta, in_act, drop, seed, out_act, res, d2s, False)
    ref = _ref_forward(x1, x2, v, scale, shift, in_act, drop, seed, out_act, res, d2s)
    scale_ = float(ref.abs().max())
    e6 = float((y6.double().cpu() - ref).abs().max())
    e32 = float((y32.double().cpu() - ref).abs().max())
    assert_close(y6, ref.float(), rtol=1e-4, atol=1e-4 * max(scale_, 1.0), name="x6 vs fp64")
    assert e6 <= 3e-6 * max(scale_, 1.0), (e6, scale_)
    assert e6 <= 5.0 * e32 + 4e-7 * max(scale_, 1.0), (e6, e32, scale_)


@pytest.mark.parametrize("cout,cin,h,w,nt,masked", [(64, 64, 8, 32, 2, False), (32, 64, 16, 32, 4, False),
                                                     (128, 64, 8, 64, 2, True), (64, 32, 4, 32, 1, True),
                                                     (256, 128, 8, 32, 2, True), (64, 64, 16, 16, 1, True),
                                                     (32, 64, 8, 16, 1, False)])
def test_x6_data_gradient_vs_fp64(cout, cin, h, w, nt, masked, scheme):
    """mode 1: dx = conv_transpose(dy [* (y > 0)], w_eff) * act'(aux) + res, against autograd in fp64."""
    ops = _ops()
    n = 2
    g_ = torch.Generator().manual_seed(cout * 7 + cin)
    v, g, bias, gamma, beta = _params(cout, cin, 9)
    wt_f, wt_d, scale, shift, invnorm, wx_f, wx_d = ops.pack_weights(v, g, bias, gamma, beta, cin, 0, 0, True)
    assert wx_d is not None
    dy = torch.randn(n, cout, h, w, generator=g_).cuda()
    yfwd = torch.randn(n, cout, h, w, generator=g_).cuda()            # stands for the forward output (ReLU mask source)
    aux = None if masked else torch.randn(n, cin, h, w, generator=g_).cuda()   # pre-activation input (ELU')
    res = torch.randn(n, cin, h, w, generator=g_).cuda()
    d = ops.ConvDesc(N=n, C1=cout, C2=0, Hs=h, Ws=w, M=cin, m_off=0, Mpad=wt_d.shape[1], Ho=h, Wo=w, KH=3, KW=3,
                     stride=1, pad=1, mode=1, in_act=0, in_slope=0.0, drop_p=0.0, drop_seed=0, out_act=0, d2s=0,
                     aux_act=0 if masked else ops.ACT_ELU, aux_slope=0.0, aux_drop_p=0.0, aux_drop_seed=0)
    dx = torch.empty(n, cin, h, w, device="cuda")
    if w % 32 and scheme != "h2":
        pytest.skip("16-wide maps: fp16 scheme only")
    ops.set_tuning("split_force_nt", nt)
    try:
        ops._call("vunet_conv2d_x6", ctypes.byref(d), ops._p(dy), None, ops._p(wx_d), None, ops._p(res), ops._p(aux),
                  ops._p(yfwd) if masked else None, ops._p(dx), _amax(ops, dy), None, ops._stream())
    finally:
        ops.set_tuning("split_force_nt", 0)
    wd = v.double().cpu() * scale.double().cpu().view(-1, 1, 1, 1)
    dyd = dy.double().cpu()
    if masked:
        dyd = dyd * (yfwd.double().cpu() > 0)
    ref = F.conv_transpose2d(dyd, wd, padding=1)
    if aux is not None:
        a = aux.double().cpu()
        ref = ref * torch.where(a > 0, torch.ones_like(a), a.exp())
    ref = ref + res.double().cpu()
    assert_close(dx, ref.float(), rtol=1e-4, atol=1e-4 * max(float(ref.abs().max()), 1.0), name="dx")
    assert float((dx.double().cpu() - ref).abs().max()) <= 3e-6 * max(float(ref.abs().max()), 1.0)


@pytest.mark.parametrize("cout,cin,h,w,nt", [(64, 64, 8, 32, 2), (64, 32, 4, 32, 1), (64, 64, 16, 16, 1)])
def test_h2_data_gradient_with_the_forward_dropout_mask(cout, cin, h, w, nt):
    """mode 1 with the layer's input dropout: dx = conv_transpose(dy, w_eff) * ELU'(aux) * keep(idx) / (1 - p) + res -- the
    16-byte epilogue's straight-line form for it (conv_common.h: store_tile_side4, FORM 3) against fp64."""
    ops = _ops()
    ops.set_conv_precision("h2")
    n, p_, seed = 2, 0.1, 77
    g_ = torch.Generator().manual_seed(cout * 5 + cin)
    v, g, bias, gamma, beta = _params(cout, cin, 9)
    wt_f, wt_d, scale, shift, invnorm, wx_f, wx_d = ops.pack_weights(v, g, bias, gamma, beta, cin, 0, 0, True)
    dy = torch.randn(n, cout, h, w, generator=g_).cuda()
    aux = torch.randn(n, cin, h, w, generator=g_).cuda()
    res = torch.randn(n, cin, h, w, generator=g_).cuda()
    d = ops.ConvDesc(N=n, C1=cout, C2=0, Hs=h, Ws=w, M=cin, m_off=0, Mpad=wt_d.shape[1], Ho=h, Wo=w, KH=3, KW=3,
                     stride=1, pad=1, mode=1, in_act=0, in_slope=0.0, drop_p=0.0, drop_seed=0, out_act=0, d2s=0,
                     aux_act=ops.ACT_ELU, aux_slope=0.0, aux_drop_p=p_, aux_drop_seed=seed)
    dx = torch.empty(n, cin, h, w, device="cuda")
    ops.set_tuning("split_force_nt", nt)
    try:
        ops._call("vunet_conv2d_x6", ctypes.byref(d), ops._p(dy), None, ops._p(wx_d), None, ops._p(res), ops._p(aux), None,
                  ops._p(dx), _amax(ops, dy), None, ops._stream())
    finally:
        ops.set_tuning("split_force_nt", 0)
    wd = v.double().cpu() * scale.double().cpu().view(-1, 1, 1, 1)
    ref = F.conv_transpose2d(dy.double().cpu(), wd, padding=1)
    a = aux.double().cpu()
    keep = dropout_keep_mask((n, cin, h, w), p_, seed).double() * float(torch.tensor(1.0 / (1.0 - p_), dtype=torch.float32))
    ref = ref * torch.where(a > 0, torch.ones_like(a), a.exp()) * keep + res.double().cpu()
    assert float((dx.double().cpu() - ref).abs().max()) <= 3e-6 * max(float(ref.abs().max()), 1.0)
    assert float((keep == 0).double().mean()) > 0.03   # the mask really dropped something


def test_x6_second_source_gradient_uses_column_offset():
    """The data gradient of the second source of a two-source layer reads the weight image at m_off = C1."""
    ops = _ops()
    n, c1, c2, cout, h, w = 2, 32, 64, 64, 8, 32
    g_ = torch.Generator().manual_seed(77)
    v, g, bias, gamma, beta = _params(cout, c1 + c2, 3)
    wt_f, wt_d, scale, shift, invnorm, wx_f, wx_d = ops.pack_weights(v, g, bias, gamma, beta, c1, c2, 0, True)
    dy = torch.randn(n, cout, h, w, generator=g_).cuda()
    d = ops.ConvDesc(N=n, C1=cout, C2=0, Hs=h, Ws=w, M=c2, m_off=c1, Mpad=wt_d.shape[1], Ho=h, Wo=w, KH=3, KW=3,
                     stride=1, pad=1, mode=1, in_act=0, in_slope=0.0, drop_p=0.0, drop_seed=0, out_act=0, d2s=0)
    dx = torch.empty(n, c2, h, w, device="cuda")
    ops._call("vunet_conv2d_x6", ctypes.byref(d), ops._p(dy), None, ops._p(wx_d), None, None, None, None, ops._p(dx),
              _amax(ops, dy), None, ops._stream())
    wd = v.double().cpu() * scale.double().cpu().view(-1, 1, 1, 1)
    ref = F.conv_transpose2d(dy.double().cpu(), wd[:, c1:], padding=1)
    assert_close(dx, ref.float(), rtol=1e-4, atol=1e-4 * max(float(ref.abs().max()), 1.0), name="dx2")


def test_x6_unsupported_geometries_are_refused_not_miscomputed():
    ops = _ops()
    lib = ops._lib.lib()

    def desc(**kw):
        base = dict(N=1, C1=32, C2=0, Hs=8, Ws=32, M=32, m_off=0, Mpad=32, Ho=8, Wo=32, KH=3, KW=3, stride=1, pad=1,
                    mode=0, in_act=0, in_slope=0.0, drop_p=0.0, drop_seed=0, out_act=0, d2s=0)
        base.update(kw)
        return ops.ConvDesc(**base)
    assert lib.vunet_conv2d_x6_supported(ctypes.byref(desc()), 0) == 1
    for bad in (dict(C1=24), dict(C1=3), dict(Ws=16, Wo=16), dict(Hs=6, Ho=6), dict(stride=2, Ho=4, Wo=16),
                dict(KH=1, KW=1, pad=0), dict(M=3), dict(in_act=4), dict(mode=1, in_act=1), dict(m_off=16)):
        assert lib.vunet_conv2d_x6_supported(ctypes.byref(desc(**bad)), 0) == 0, bad
    x = torch.randn(1, 24, 8, 32, device="cuda")
    y = torch.empty(1, 32, 8, 32, device="cuda")
    wx = torch.zeros(1024, device="cuda", dtype=torch.int32)
    rc = lib.vunet_conv2d_x6(ctypes.byref(desc(C1=24)), ops._p(x), None, ops._p(wx), None, None, None, None, ops._p(y),
                             _amax(ops, x), None, ops._stream())
    assert rc == -3   # VUNET_ERR_UNSUPPORTED


def test_fused_conv_takes_the_x6_path_and_matches_the_f32_path(scheme):
    """Through ops.fused_conv + autograd at a size the dispatcher routes to the split kernels: forward, input and
    parameter gradients agree with the fp32-MFMA path to fp32 accuracy, and the profiler sees the scheme's kernel."""
    ops = _ops()
    fam_name = "conv_x6_kernel" if scheme == "x6" else "conv_h2_kernel"
    from behavior_driven_video_synthesis_amd.lib.modules import VunetRNB
    torch.manual_seed(5)
    blk = VunetRNB(64, a_channels=64, residual=True, dropout_prob=0.05).cuda().train()
    x = torch.randn(4, 64, 64, 64, device="cuda")
    a = torch.randn(4, 64, 64, 64, device="cuda")
    wgt = torch.randn(4, 64, 64, 64, device="cuda")
    outs = {}
    for mode in (scheme, "f32"):
        ops.set_conv_precision(mode)
        ops.set_dropout_seed(31)
        xi, ai = x.clone().requires_grad_(True), a.clone().requires_grad_(True)
        blk.zero_grad()
        ops.profile_start()
        y = blk(xi, ai)
        (y * wgt).sum().backward()
        fam = ops.profile_stop(by_kernel=True)
        outs[mode] = (y.detach(), xi.grad, ai.grad, {k: p.grad.clone() for k, p in blk.named_parameters()}, fam)
    assert any(k.startswith(fam_name) for k in outs[scheme][4]), list(outs[scheme][4])
    assert not any(k.startswith(("conv_x6_kernel", "conv_h2_kernel")) for k in outs["f32"][4]), list(outs["f32"][4])
    y6, gx6, ga6, gp6, _ = outs[scheme]
    y3, gx3, ga3, gp3, _ = outs["f32"]
    assert_close(y6, y3, rtol=2e-5, atol=2e-5 * float(y3.abs().max()), name="y")
    assert_close(gx6, gx3, rtol=2e-5, atol=2e-5 * float(gx3.abs().max()), name="gx")
    assert_close(ga6, ga3, rtol=2e-5, atol=2e-5 * float(ga3.abs().max()), name="ga")
    for k in gp3:
        assert_close(gp6[k], gp3[k], rtol=1e-4, atol=1e-4 * float(gp3[k].abs().max()) + 1e-7, name=k)


@pytest.mark.parametrize("n,c1,c2,cout,h,w,in_act,drop", [
    (2, 32, 0, 32, 8, 32, 0, 0.0),      # MTW 1: four waves fold one m-tile
    (2, 32, 32, 32, 8, 64, 1, 0.05),    # two sources, ELU + dropout prologue, two column tiles (halo columns live)
    (1, 64, 0, 64, 12, 32, 1, 0.0),     # MTW 2, three row tiles (top / interior / bottom halo rows)
    (3, 64, 64, 128, 4, 96, 1, 0.1),    # MTW 2, two co-blocks, three column tiles
    (2, 32, 0, 96, 8, 32, 0, 0.0),      # Cout = 96: not a multiple of 64 -> MTW 1, three co-blocks
])
@pytest.mark.parametrize("wide", [False, True])
def test_x6_weight_gradient_vs_fp64(n, c1, c2, cout, h, w, in_act, drop, wide, scheme):
    """conv_wgrad_x6_kernel / conv_wgrad_h2_kernel: slabs summed by vunet_weightnorm_bwd (kind 1: plain dW) against fp64
    autograd, beside the fp32-MFMA weight-gradient kernel on the same problem.  ``wide``: the magnitudes of x and dy fall
    by 2^-0.75 per pixel column (2^-23 across the tile) -- the data a tensor-wide scale is worst for."""
    ops = _ops()
    g_ = torch.Generator().manual_seed(n * 1000 + c1 + cout)
    col = torch.pow(2.0, -0.75 * (torch.arange(w, dtype=torch.float32) % 32)).view(1, 1, 1, w) if wide else 1.0
    x1 = (torch.randn(n, c1, h, w, generator=g_) * col).cuda()
    x2 = (torch.randn(n, c2, h, w, generator=g_) * col).cuda() if c2 else None
    dy = (torch.randn(n, cout, h, w, generator=g_) * col).cuda()
    seed = 0xBEEF
    ctot, ktot = c1 + c2, 9 * (c1 + c2)

    def run(flags):
        wd = ops.WgradDesc(N=n, C1=c1, C2=c2, Hs=h, Ws=w, Cout=cout, Ho=h, Wo=w, KH=3, KW=3, stride=1, pad=1,
                           in_act=in_act, in_slope=0.0, drop_p=drop, drop_seed=seed, nsplit=1, flags=flags)
        buf = ctypes.create_string_buffer(96)
        ops._call("vunet_conv2d_wgrad_variant", ctypes.byref(wd), buf, 96)
        ns = ops._lib.lib().vunet_conv2d_wgrad_nsplit(ctypes.byref(wd))
        assert ns >= 1
        wd.nsplit = ns
        cp = ops._r32(cout)
        slabs = torch.full((ns * cp * ktot + ns * cp,), float("nan"), device="cuda")
        dshift = slabs[ns * cp * ktot:]
        amx = ops.absmax_partials(x1, x2) if flags == 2 else None
        amd = ops.absmax_partials(dy) if flags == 2 else None
        ops._call("vunet_conv2d_wgrad", ctypes.byref(wd), ops._p(x1), ops._p(x2), ops._p(dy), ops._p(slabs),
                  ops._p(dshift), ops._p(amx), ops._p(amd), ops._stream())
        dw = torch.empty(cout, ctot, 3, 3, device="cuda")
        db = torch.empty(cout, device="cuda")
        v = torch.zeros(cout, ctot, 3, 3, device="cuda")
        work = torch.empty(cout * (ktot + 1), device="cuda")
        wn = ops.WnDesc(cout, c1, c2, 3, 3, 1, 0)
        ops._call("vunet_weightnorm_bwd", ctypes.byref(wn), ops._p(slabs), ops._p(dshift), ns, ops._p(v), None, None,
                  None, None, ops._p(dw), None, ops._p(db), None, None, ops._p(work), 0, ops._stream())
        torch.cuda.synchronize()
        return dw, db, buf.value.decode()

    dw6, db6, name6 = run(2 if scheme == "h2" else 0)
    dw3, db3, name3 = run(1)
    assert name6.startswith("conv_wgrad_%s_kernel" % scheme) and not name3.startswith("conv_wgrad_%s_kernel" % scheme), (name6, name3)
    # fp64 reference: the same prologue as the forward, then autograd of the convolution w.r.t. the weight
    xs = []
    for i, x in enumerate((x1, x2)):
        if x is None:
            continue
        t = x.double().cpu()
        if in_act == ops.ACT_ELU:
            t = F.elu(t)
        if drop > 0:
            s = seed if i == 0 else (seed + ops.SEED2_OFFSET) & 0xFFFFFFFF
            t = t * dropout_keep_mask(tuple(x.shape), drop, s).double() * float(torch.tensor(1.0 / (1.0 - drop),
                                                                                             dtype=torch.float32))
        xs.append(t)
    xin = torch.cat(xs, dim=1)
    wz = torch.zeros(cout, ctot, 3, 3, dtype=torch.float64, requires_grad=True)
    (F.conv2d(xin, wz, padding=1) * dy.double().cpu()).sum().backward()
    ref = wz.grad
    scale_ = float(ref.abs().max())
    e6 = float((dw6.double().cpu() - ref).abs().max())
    e3 = float((dw3.double().cpu() - ref).abs().max())
    assert e6 <= 3e-6 * scale_, (e6, scale_)
    assert e6 <= 5.0 * e3 + 4e-7 * scale_, (e6, e3, scale_)
    refb = dy.double().cpu().sum(dim=(0, 2, 3))
    assert_close(db6, refb.float(), rtol=1e-5, atol=1e-5 * float(refb.abs().max()), name="dshift")


@pytest.mark.parametrize("cout,cin,h,w,nt,with_aux", [(64, 32, 8, 32, 2, False), (64, 32, 16, 64, 4, True),
                                                       (128, 64, 8, 32, 2, False), (128, 128, 4, 64, 1, True)])
def test_x6_stride2_data_gradient_vs_fp64(cout, cin, h, w, nt, with_aux):
    """Data gradient of the stride-2 Downsample conv (lib/modules.py:152-158) as four output-parity launches of the
    split-bf16 kernel over the dy map [h, w] -> dx [2h, 2w], against autograd's transposed convolution in fp64."""
    ops = _ops()
    n = 2
    g_ = torch.Generator().manual_seed(cout + 3 * cin + h)
    v, g, bias, gamma, beta = _params(cout, cin, 13)
    wt_f, wt_d, scale, shift, invnorm, wx_f, wx_d = ops.pack_weights(v, g, bias, gamma, beta, cin, 0, 0, True)
    dy = torch.randn(n, cout, h, w, generator=g_).cuda()
    aux = torch.randn(n, cin, 2 * h, 2 * w, generator=g_).cuda() if with_aux else None
    res = torch.randn(n, cin, 2 * h, 2 * w, generator=g_).cuda() if with_aux else None
    d = ops.ConvDesc(N=n, C1=cout, C2=0, Hs=h, Ws=w, M=cin, m_off=0, Mpad=wt_d.shape[1], Ho=2 * h, Wo=2 * w, KH=3, KW=3,
                     stride=2, pad=1, mode=1, in_act=0, in_slope=0.0, drop_p=0.0, drop_seed=0, out_act=0, d2s=0,
                     aux_act=ops.ACT_ELU if with_aux else 0, aux_slope=0.0, aux_drop_p=0.0, aux_drop_seed=0)
    assert ops._lib.lib().vunet_conv2d_x6_supported(ctypes.byref(d), 0) == 1
    wd = v.double().cpu() * scale.double().cpu().view(-1, 1, 1, 1)
    ref = F.conv_transpose2d(dy.double().cpu(), wd, stride=2, padding=1, output_padding=1)
    if with_aux:
        a = aux.double().cpu()
        ref = ref * torch.where(a > 0, torch.ones_like(a), a.exp()) + res.double().cpu()
    # the fp16 scheme has two forms: all four parities fused in one launch (default) and one launch per parity
    for per_parity in (0, 1):
        dx = torch.full((n, cin, 2 * h, 2 * w), float("nan"), device="cuda")     # every pixel must be written by some parity
        amax_out = torch.zeros(1024, device="cuda")
        ops.set_tuning("split_force_nt", nt)
        ops.set_tuning("parity_launches", per_parity)
        try:
            am = _amax(ops, dy)
            ops._call("vunet_conv2d_x6", ctypes.byref(d), ops._p(dy), None, ops._p(wx_d), None, ops._p(res), ops._p(aux), None,
                      ops._p(dx), am, ops._p(amax_out) if am is not None else None, ops._stream())
        finally:
            ops.set_tuning("split_force_nt", 0)
            ops.set_tuning("parity_launches", 0)
        assert torch.isfinite(dx).all()
        assert_close(dx, ref.float(), rtol=1e-4, atol=1e-4 * max(float(ref.abs().max()), 1.0), name="dx")
        assert float((dx.double().cpu() - ref).abs().max()) <= 3e-6 * max(float(ref.abs().max()), 1.0)
        if am is not None:   # the published maxima are those of what was stored
            assert float(amax_out.max()) == float(dx.abs().max())
    # and the fp32 per-parity gather path on the same problem
    dx32 = torch.empty_like(dx)
    ops._call("vunet_conv2d_gather", ctypes.byref(d), ops._p(dy), None, ops._p(wt_d), None, ops._p(res), ops._p(aux),
              ops._p(dx32), ops._stream())
    assert_close(dx, dx32, rtol=2e-5, atol=2e-5 * max(float(ref.abs().max()), 1.0), name="x6 vs fp32 path")


def test_absmax_partials_and_the_h2_weight_image(scheme):
    """The two data-dependent inputs of the fp16 scheme: vunet_absmax_partials (512 partial |x| maxima per source, NaNs
    ignored, odd sizes and unaligned tails) and the weight image's header exponent (the layer's largest |w_eff| lands in
    [2^13, 2^14)) with planes that reassemble the scaled weight to 2^-22."""
    if scheme != "h2":
        pytest.skip("fp16 scheme only")
    ops = _ops()
    g_ = torch.Generator().manual_seed(3)
    for n1, n2 in ((1000003, 0), (4096, 77), (5, 5)):
        x1 = torch.randn(n1, generator=g_).cuda() * 3.0
        x2 = (torch.randn(n2, generator=g_).cuda() * 0.01) if n2 else None
        x1[n1 // 2] = float("nan")                      # ignored: NaNs travel with the data, not with the scale
        p = ops.absmax_partials(x1, x2)
        want1 = float(torch.nan_to_num(x1, nan=0.0).abs().max())
        assert float(p[:512].max()) == want1
        assert float(p[512:].max()) == (float(x2.abs().max()) if n2 else 0.0)
    cout, cin = 64, 32
    v, g, bias, gamma, beta = _params(cout, cin, 21)
    wt_f, wt_d, scale, shift, invnorm, wx_f, wx_d = ops.pack_weights(v, g, bias, gamma, beta, cin, 0, 0, True)
    w_eff = (v * scale.view(-1, 1, 1, 1)).cpu()
    wmax = float(w_eff.abs().max())
    for wx in (wx_f, wx_d):
        ew = int(wx[0].item())
        assert 2.0 ** 13 <= wmax * 2.0 ** ew < 2.0 ** 14, (wmax, ew)
    # forward image: [chunk][kh][m-tile][kw][plane][k-half][32] units of 8 fp16 behind the 16-byte header
    img = wx_f[4:].view(torch.float16).view(cin // 16, 3, ops.x6_mtiles(cout), 3, 2, 2, 32, 8).float().cpu()
    ew = int(wx_f[0].item())
    rec = (img[:, :, :, :, 0] + img[:, :, :, :, 1] / 2048.0) / 2.0 ** ew          # [ch][kh][mt][kw][half][j][e]
    for (co, ci, kh, kw) in ((0, 0, 0, 0), (63, 31, 2, 2), (17, 9, 1, 2), (40, 24, 0, 1)):
        got = float(rec[ci // 16, kh, co // 32, kw, (ci % 16) // 8, co % 32, ci % 8])
        want = float(w_eff[co, ci, kh, kw])
        assert abs(got - want) <= 2.0 ** -21 * abs(want) + 1e-12, (co, ci, kh, kw, got, want)
    assert float(img[:, :, 2:].abs().max()) == 0.0     # the padding m-tiles are zeros


def test_producer_maxima_tag_equals_a_pass_over_the_tensor(scheme):
    """The h2 kernel's epilogue publishes max|y| of what it stores (vunet_conv2d: amax_out) and ops tags y with it: the
    tag must equal what vunet_absmax_partials finds in y (forward with residual, data gradient with act' and residual,
    through max-pool), a single-source consumer must use it (no absmax launch), and an in-place change must void it."""
    if scheme != "h2":
        pytest.skip("fp16 scheme only")
    ops = _ops()
    from behavior_driven_video_synthesis_amd.lib.modules import NormConv2d
    torch.manual_seed(11)
    conv1 = NormConv2d(64, 64, 3, padding=1).cuda()
    conv2 = NormConv2d(64, 64, 3, padding=1).cuda()
    x = (torch.randn(8, 64, 64, 64, device="cuda") * 3.0).requires_grad_(True)
    y = conv1(x)
    tag = ops._tagged_amax(y)
    assert tag is not None and float(tag.max()) == float(y.detach().abs().max()) and float(tag[512:].max()) == 0.0
    p = ops.MaxPool2.apply(y)
    assert ops._tagged_amax(p) is tag                                   # a bound for the pooled tensor, passed on
    calls = []
    orig = ops.absmax_partials
    ops.absmax_partials = lambda *a: (calls.append(1), orig(*a))[1]
    try:
        z = conv2(y)                                                     # single tagged source: no pass of its own
        assert not calls
        z.square().mean().backward()                                     # |dy| of conv2 comes from autograd: one pass
        n_bwd = len(calls)
        y2 = y.detach().clone()
        assert ops._tagged_amax(y2) is None
        y3 = y.detach()
        y3.mul_(2.0)                                                     # in place: the tag (same storage) is void
        assert ops._tagged_amax(y) is None
        conv2(y3)
        assert len(calls) == n_bwd + 1
    finally:
        ops.absmax_partials = orig
    assert x.grad is not None and torch.isfinite(x.grad).all()


@pytest.mark.parametrize("mag", [0.0, 1e-30, 1e-12, 1e12, 1e30])
def test_h2_scale_handles_extreme_magnitudes(mag, scheme):
    """The power-of-two scale of the fp16 scheme at the ends of the fp32 range: an all-zero input gives exactly the shift,
    tensors around 1e-30 / 1e+30 are scaled into fp16's range and back without loss (the exponent is clamped to +-60 per
    operand: beyond that the relative accuracy degrades gracefully, it never overflows), NaN and Inf propagate."""
    if scheme != "h2":
        pytest.skip("fp16 scheme only")
    ops = _ops()
    n, c, h, w, cout = 2, 32, 8, 32, 32
    g_ = torch.Generator().manual_seed(5)
    x = (torch.randn(n, c, h, w, generator=g_) * mag).cuda()
    v, g, bias, gamma, beta = _params(cout, c, 17)
    y, scale, shift, _ = _run_forward(ops, x, None, v, g, bias, gamma, beta, 0, 0.0, 0, 0, None, False, True, 2)
    ref = _ref_forward(x, None, v, scale, shift, 0, 0.0, 0, 0, None, False)
    conv_part = (ref - shift.double().cpu().view(1, -1, 1, 1))
    tol = 3e-6 * float(conv_part.abs().max()) + 2e-7 * float(shift.abs().max())   # fp32 rounding of  conv + shift
    assert torch.isfinite(y).all()
    assert float((y.double().cpu() - ref).abs().max()) <= tol, (mag, float((y.double().cpu() - ref).abs().max()), tol)
    if mag == 0.0:
        assert torch.equal(y, shift.view(1, -1, 1, 1).expand_as(y))
    if mag == 1e12:   # NaN / Inf travel with the data (the |x| maxima ignore NaN; an Inf maximum leaves the data unscaled)
        xn = x.clone()
        xn[0, 3, 4, 5] = float("nan")
        yn, _, _, _ = _run_forward(ops, xn, None, v, g, bias, gamma, beta, 0, 0.0, 0, 0, None, False, True, 2)
        assert torch.isnan(yn[0, :, 3:6, 4:7]).all() and torch.isfinite(yn[1]).all()
        assert torch.isfinite(yn[0, :, :, 16:]).all()


@pytest.mark.parametrize("n,cin,cout,ho,wo,relu", [(2, 32, 64, 32, 64, False), (4, 64, 128, 16, 32, False),
                                                     (33, 16, 32, 8, 32, True), (2, 128, 128, 32, 32, False)])
def test_h2_stride2_forward_vs_fp64(n, cin, cout, ho, wo, relu):
    """Forward of the stride-2 Downsample conv (lib/modules.py:148-161) on the fp16 scheme's parity-plane kernel
    (csrc/conv_h2_s2.hip): input [2 ho, 2 wo] -> output [ho, wo], against an fp64 convolution; the same problem on the
    fp32-input MFMA kernel it replaces (tuning knob) must agree to fp32 accuracy; published maxima = those of y."""
    ops = _ops()
    if ops.conv_precision() != "h2":
        pytest.skip("fp16 scheme only")
    g_ = torch.Generator().manual_seed(cin + cout + ho)
    v, g, bias, gamma, beta = _params(cout, cin, 5)
    wt_f, wt_d, scale, shift, invnorm, wx_f, wx_d = ops.pack_weights(v, g, bias, gamma, beta, cin, 0, 0, False)
    x = torch.randn(n, cin, 2 * ho, 2 * wo, generator=g_).cuda()
    d = ops.ConvDesc(N=n, C1=cin, C2=0, Hs=2 * ho, Ws=2 * wo, M=cout, m_off=0, Mpad=wt_f.shape[1], Ho=ho, Wo=wo, KH=3, KW=3,
                     stride=2, pad=1, mode=0, in_act=0, in_slope=0.0, drop_p=0.0, drop_seed=0,
                     out_act=ops.ACT_RELU if relu else 0, d2s=0)
    assert ops._lib.lib().vunet_conv2d_wants_split(ctypes.byref(d), 0, 0, 0, 2) == 1
    buf = ctypes.create_string_buffer(96)
    ops._call("vunet_conv2d_variant", ctypes.byref(d), 0, 2, 0, buf, 96)
    assert buf.value.decode().startswith("conv_h2_s2_kernel<"), buf.value
    y = torch.full((n, cout, ho, wo), float("nan"), device="cuda")
    amax_out = torch.zeros(1024, device="cuda")
    ops._call("vunet_conv2d", ctypes.byref(d), ops._p(x), None, ops._p(wt_f), ops._p(wx_f), ops._p(shift), None, None,
              ops._p(y), _amax(ops, x), ops._p(amax_out), ops._stream())
    w64 = v.double().cpu() * scale.double().cpu().view(-1, 1, 1, 1)
    ref = F.conv2d(x.double().cpu(), w64, None, stride=2, padding=1) + shift.double().cpu().view(1, -1, 1, 1)
    if relu:
        ref = torch.relu(ref)
    s_ = max(float(ref.abs().max()), 1.0)
    assert torch.isfinite(y).all()
    assert float((y.double().cpu() - ref).abs().max()) <= 3e-6 * s_
    assert float(amax_out.max()) == float(y.abs().max())
    ops.set_tuning("s2_fwd_f32", 1)
    try:
        ops._call("vunet_conv2d_variant", ctypes.byref(d), 0, 2, 0, buf, 96)
        assert not buf.value.decode().startswith("conv_h2_s2_kernel"), buf.value
        y32 = torch.empty_like(y)
        ops._call("vunet_conv2d", ctypes.byref(d), ops._p(x), None, ops._p(wt_f), ops._p(wx_f), ops._p(shift), None, None,
                  ops._p(y32), None, None, ops._stream())
    finally:
        ops.set_tuning("s2_fwd_f32", 0)
    assert_close(y, y32, rtol=2e-5, atol=2e-5 * s_, name="h2 stride-2 forward vs the fp32 MFMA kernel")
