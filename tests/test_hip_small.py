"""-m gpu: the small-map K-split convolution kernel (csrc/conv_h2_small.hip) against an fp64 convolution.

The kernel serves the VUnet's bottleneck (models/vunets.py:520-597, :264-424: 4 x 4 ... 16 x 16 maps at 128 channels) and
the Downsample convolutions that lead there (lib/modules.py:148-161): forward at stride 1 and 2, data gradient at stride
1 and of the stride-2 convolution (output parity classes), with every prologue / epilogue of the fused layer.  Same
bounds as tests/test_hip_x6.py: 3e-6 of the output scale against fp64 (fp32-level accuracy from two fp16 terms / three
products), and the 1e-4 parity tolerance.  Ragged cases: a batch whose pixels do not fill the last 32-pixel tile, tiles
spanning two and three images, odd map sizes.
"""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from hip_parity_utils import assert_close, dropout_keep_mask

pytestmark = pytest.mark.gpu


def _ops():
    from behavior_driven_video_synthesis_amd import ops
    return ops


@pytest.fixture(autouse=True)
def h2_scheme_and_small_kernel():
    ops = _ops()
    before = ops.conv_precision()
    ops.set_conv_precision("h2")
    ops.set_tuning("force_small", 1)
    yield
    ops.set_tuning("force_small", 0)
    ops.set_conv_precision(before)


def _params(cout, cin, seed):
    g_ = torch.Generator().manual_seed(seed)
    v = (torch.randn(cout, cin, 3, 3, generator=g_) * 0.2).cuda()
    g = (torch.rand(cout, 1, 1, 1, generator=g_) + 0.5).cuda()
    bias = (torch.randn(cout, generator=g_) * 0.1).cuda()
    gamma = (1.0 + 0.3 * torch.randn(1, cout, 1, 1, generator=g_)).cuda()
    beta = (0.2 * torch.randn(1, cout, 1, 1, generator=g_)).cuda()
    return v, g, bias, gamma, beta


def _kernel_name(ops, d, has_aux=False):
    buf = ctypes.create_string_buffer(96)
    ops._call("vunet_conv2d_variant", ctypes.byref(d), int(has_aux), 2, 0, buf, 96)
    return buf.value.decode()


# (n, c1, c2, cout, h, w, stride, in_act, drop, out_act, with_res, d2s)
FWD = [
    (16, 128, 0, 128, 4, 4, 1, 1, 0.05, 0, True, False),    # the bottleneck RNB conv at 4 x 4: tiles of two images
    (16, 128, 128, 128, 8, 8, 1, 1, 0.05, 0, True, False),  # two sources (skip read), 8 x 8: 16 chunks over 8 waves
    (4, 128, 0, 128, 16, 16, 1, 1, 0.0, 0, True, False),    # 16 x 16: two rows per tile
    (2, 128, 0, 512, 4, 4, 1, 0, 0.0, 0, False, True),      # sub-pixel up-conv through the depth-to-space store
    (2, 64, 0, 64, 8, 8, 1, 0, 0.0, 3, False, False),       # sigmoid epilogue (the log-std head)
    (1, 32, 0, 32, 4, 4, 1, 0, 0.0, 0, False, False),       # 16 pixels in all: half a tile, waves without a chunk
    (3, 48, 16, 96, 4, 4, 1, 1, 0.1, 0, True, False),       # 48 pixels: a tile over two images and a ragged last tile
    (5, 32, 0, 64, 7, 7, 1, 1, 0.0, 0, True, False),        # odd map: tiles start anywhere, up to two images each
    (16, 128, 0, 128, 8, 8, 2, 0, 0.0, 0, False, False),    # Downsample 8 -> 4
    (4, 64, 0, 128, 16, 16, 2, 0, 0.0, 0, False, False),    # Downsample 16 -> 8
    (2, 32, 0, 64, 32, 32, 2, 0, 0.0, 0, False, False),     # Downsample 32 -> 16: five staging rounds
    (3, 16, 0, 32, 5, 7, 2, 1, 0.0, 0, False, False),       # stride 2 on an odd, non-square map (tiles over 3 images)
]


@pytest.mark.parametrize("case", FWD)
def test_small_forward_vs_fp64(case):
    ops = _ops()
    n, c1, c2, cout, h, w, stride, in_act, drop, out_act, with_res, d2s = case
    g_ = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x1 = torch.randn(n, c1, h, w, generator=g_).cuda()
    x2 = torch.randn(n, c2, h, w, generator=g_).cuda() if c2 else None
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    res = None
    if with_res:
        res = torch.randn((n, cout // 4, 2 * ho, 2 * wo) if d2s else (n, cout, ho, wo), generator=g_).cuda()
    v, g, bias, gamma, beta = _params(cout, c1 + c2, 5)
    wt_f, wt_d, scale, shift, invnorm, wx_f, wx_d = ops.pack_weights(v, g, bias, gamma, beta, c1, c2, 0, True)
    seed = 0xBEEF01
    y = torch.empty((n, cout // 4, 2 * ho, 2 * wo) if d2s else (n, cout, ho, wo), device="cuda")
    d = ops.ConvDesc(N=n, C1=c1, C2=c2, Hs=h, Ws=w, M=cout, m_off=0, Mpad=wt_f.shape[1], Ho=ho, Wo=wo, KH=3, KW=3,
                     stride=stride, pad=1, mode=0, in_act=in_act, in_slope=0.0, drop_p=drop, drop_seed=seed,
                     out_act=out_act, d2s=int(d2s))
    amax_out = torch.zeros(1024, device="cuda")
    ops._call("vunet_conv2d_x6", ctypes.byref(d), ops._p(x1), ops._p(x2), ops._p(wx_f), ops._p(shift), ops._p(res), None,
              None, ops._p(y), ops._p(ops.absmax_partials(x1, x2)), ops._p(amax_out), ops._stream())
    torch.cuda.synchronize()
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
    wd = v.double().cpu() * scale.double().cpu().view(-1, 1, 1, 1)
    ref = F.conv2d(torch.cat(xs, dim=1), wd, None, stride=stride, padding=1) + shift.double().cpu().view(1, -1, 1, 1)
    if out_act == ops.ACT_SIGMOID:
        ref = torch.sigmoid(ref)
    if d2s:
        from oracle import vunet_oracle as O
        ref = O.depth_to_space(ref)
    if res is not None:
        ref = ref + res.double().cpu()
    sc = max(float(ref.abs().max()), 1.0)
    assert_close(y, ref.float(), rtol=1e-4, atol=1e-4 * sc, name="y")
    assert float((y.double().cpu() - ref).abs().max()) <= 3e-6 * sc
    assert float(amax_out.max()) == float(y.abs().max())   # the published |y| maxima (also through the depth-to-space store)


def test_dispatcher_routes_the_bottleneck_layers_to_the_small_kernel():
    """vunet_conv2d's own choice (no forcing) for the bs-16 layers of BASELINE config 2 the row-tiled kernels cannot fill
    the chip with."""
    ops = _ops()
    ops.set_tuning("force_small", 0)
    for (c1, c2, cout, hs, stride, mode) in [(128, 0, 128, 4, 1, 0), (128, 128, 128, 8, 1, 0), (128, 0, 128, 16, 1, 0),
                                              (128, 0, 128, 16, 2, 0), (128, 0, 128, 32, 2, 0), (128, 0, 128, 8, 1, 1),
                                              (128, 0, 128, 4, 2, 1), (128, 0, 128, 16, 2, 1), (512, 0, 128, 8, 1, 1)]:
        ho = hs if stride == 1 else (hs // 2 if mode == 0 else hs * 2)
        d = ops.ConvDesc(N=16, C1=c1, C2=c2, Hs=hs, Ws=hs, M=cout, m_off=0, Mpad=128, Ho=ho, Wo=ho, KH=3, KW=3,
                         stride=stride, pad=1, mode=mode, in_act=0, in_slope=0.0, drop_p=0.0, drop_seed=0, out_act=0, d2s=0)
        assert ops._lib.lib().vunet_conv2d_wants_split(ctypes.byref(d), 0, 0, 0, 2) == 1, (c1, c2, cout, hs, stride, mode)
        assert _kernel_name(ops, d).startswith("conv_h2_small_kernel"), (_kernel_name(ops, d), hs, stride, mode)
    # big maps stay on the row-tiled kernel
    d = ops.ConvDesc(N=16, C1=128, C2=0, Hs=32, Ws=32, M=128, m_off=0, Mpad=128, Ho=32, Wo=32, KH=3, KW=3, stride=1, pad=1,
                     mode=0, in_act=0, in_slope=0.0, drop_p=0.0, drop_seed=0, out_act=0, d2s=0)
    assert _kernel_name(ops, d).startswith("conv_h2_kernel")


# (n, cout(forward), cin(forward), h, w of dy, stride, aux_act, aux_drop, with_res, second source offset)
DGRAD = [
    (16, 128, 128, 4, 4, 1, 1, 0.05, True, 0),     # data gradient of the 4 x 4 RNB conv: ELU' * dropout mask, + residual
    (4, 128, 128, 16, 16, 1, 1, 0.0, True, 0),
    (2, 512, 128, 8, 8, 1, 0, 0.0, False, 0),      # through the sub-pixel up-conv (dy arrives space-to-depth'ed): 32 chunks
    (3, 32, 96, 4, 4, 1, 1, 0.0, False, 32),       # second source of a two-source layer: weight columns from m_off
    (16, 128, 128, 4, 4, 2, 0, 0.0, False, 0),     # Downsample 8 -> 4 backwards: four parity classes, 2 images per tile
    (4, 128, 64, 8, 8, 2, 0, 0.0, False, 0),       # dy 8 x 8 -> dx 16 x 16
    (2, 64, 32, 16, 16, 2, 0, 0.0, False, 0),      # dy 16 x 16 -> dx 32 x 32
]


@pytest.mark.parametrize("case", DGRAD)
def test_small_data_gradient_vs_fp64(case):
    ops = _ops()
    n, cout, cin, h, w, stride, aux_act, aux_drop, with_res, m_off = case
    g_ = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    ctot = cin + m_off
    v, g, bias, gamma, beta = _params(cout, ctot, 9)
    wt_f, wt_d, scale, shift, invnorm, wx_f, wx_d = ops.pack_weights(v, g, bias, gamma, beta, m_off if m_off else ctot,
                                                                      cin if m_off else 0, 0, True)
    ho, wo = h * stride, w * stride
    dy = torch.randn(n, cout, h, w, generator=g_).cuda()
    aux = torch.randn(n, cin, ho, wo, generator=g_).cuda() if (aux_act or aux_drop) else None
    res = torch.randn(n, cin, ho, wo, generator=g_).cuda() if with_res else None
    seed = 0xD06F00D
    d = ops.ConvDesc(N=n, C1=cout, C2=0, Hs=h, Ws=w, M=cin, m_off=m_off, Mpad=wt_d.shape[1], Ho=ho, Wo=wo, KH=3, KW=3,
                     stride=stride, pad=1, mode=1, in_act=0, in_slope=0.0, drop_p=0.0, drop_seed=0, out_act=0, d2s=0,
                     aux_act=aux_act, aux_slope=0.0, aux_drop_p=aux_drop, aux_drop_seed=seed)
    dx = torch.empty(n, cin, ho, wo, device="cuda")
    ops._call("vunet_conv2d_x6", ctypes.byref(d), ops._p(dy), None, ops._p(wx_d), None, ops._p(res), ops._p(aux), None,
              ops._p(dx), ops._p(ops.absmax_partials(dy)), None, ops._stream())
    torch.cuda.synchronize()
    wd = (v.double().cpu() * scale.double().cpu().view(-1, 1, 1, 1))[:, m_off:m_off + cin]
    ref = F.conv_transpose2d(dy.double().cpu(), wd, stride=stride, padding=1, output_padding=stride - 1)
    if aux is not None:
        a = aux.double().cpu()
        gr = torch.where(a > 0, torch.ones_like(a), a.exp()) if aux_act == ops.ACT_ELU else torch.ones_like(a)
        if aux_drop > 0:
            gr = gr * dropout_keep_mask(tuple(aux.shape), aux_drop, seed).double() * float(
                torch.tensor(1.0 / (1.0 - aux_drop), dtype=torch.float32))
        ref = ref * gr
    if res is not None:
        ref = ref + res.double().cpu()
    sc = max(float(ref.abs().max()), 1.0)
    assert_close(dx, ref.float(), rtol=1e-4, atol=1e-4 * sc, name="dx")
    assert float((dx.double().cpu() - ref).abs().max()) <= 3e-6 * sc


# ---- weight gradient of the same layers: conv_wgrad_direct_kernel (csrc/conv_wgrad_direct.hip)
# (n, c1, c2, cout, hs, ws, k, stride, in_act, drop)
WGRAD = [
    (16, 128, 0, 128, 4, 4, 3, 1, 1, 0.05),     # 4 x 4 bottleneck conv: octets of two rows
    (4, 128, 128, 128, 8, 8, 3, 1, 1, 0.05),    # two sources at 8 x 8
    (2, 64, 0, 96, 16, 16, 3, 1, 1, 0.0),       # 16 x 16, Cout = 96 (three co tiles, the last unit of a workgroup idle)
    (2, 32, 0, 32, 6, 24, 3, 1, 0, 0.0),        # non-square, three octets per row: left / interior / right halo columns
    (16, 128, 0, 128, 8, 8, 3, 2, 0, 0.0),      # Downsample 8 -> 4: 4-wide output, five input rows per octet
    (4, 64, 0, 128, 16, 16, 3, 2, 0, 0.0),      # Downsample 16 -> 8
    (2, 32, 0, 64, 64, 64, 3, 2, 0, 0.0),       # Downsample on a large map (the 256 -> 128 layer at reduced size)
    (2, 32, 0, 32, 7, 16, 3, 2, 1, 0.0),        # odd input height: the last tap row falls outside
    (2, 32, 0, 32, 32, 32, 1, 1, 1, 0.0),       # 1 x 1 nin, ELU prologue
    (3, 64, 32, 128, 8, 8, 1, 1, 1, 0.05),      # 1 x 1, two sources, dropout
    (2, 3, 0, 32, 16, 16, 1, 1, 0, 0.0),        # 1 x 1 on the 3-channel input (columns past the end are zero)
    (2, 32, 0, 16, 16, 16, 3, 1, 0, 0.0),       # 16 output channels: half of the co tile is zero rows
]


@pytest.mark.parametrize("rowsplit", [0, 1, 4], ids=["rowsplit_auto", "rowsplit_off", "rowsplit_stride2_only"])
@pytest.mark.parametrize("case", WGRAD)
def test_direct_weight_gradient_vs_fp64(case, rowsplit):
    """(rowsplit: the kernel-row split of the direct kernel -- one kernel row per wave -- by default on every 3x3 layer of
    maps >= 8 wide, switched off / narrowed to the stride-2 layers by the tuning knob)"""
    ops = _ops()
    ops.set_tuning("wgrad_rowsplit", rowsplit)
    n, c1, c2, cout, hs, ws, k, stride, in_act, drop = case
    pad = 1 if k == 3 else 0
    ho, wo = (hs + 2 * pad - k) // stride + 1, (ws + 2 * pad - k) // stride + 1
    g_ = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x1 = torch.randn(n, c1, hs, ws, generator=g_).cuda()
    x2 = torch.randn(n, c2, hs, ws, generator=g_).cuda() if c2 else None
    dy = torch.randn(n, cout, ho, wo, generator=g_).cuda()
    seed = 0xFACE
    ctot, ktot = c1 + c2, k * k * (c1 + c2)

    def run(flags):
        wd = ops.WgradDesc(N=n, C1=c1, C2=c2, Hs=hs, Ws=ws, Cout=cout, Ho=ho, Wo=wo, KH=k, KW=k, stride=stride, pad=pad,
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
        dw = torch.empty(cout, ctot, k, k, device="cuda")
        db = torch.empty(cout, device="cuda")
        v = torch.zeros(cout, ctot, k, k, device="cuda")
        work = torch.empty(cout * (ktot + 1), device="cuda")
        wn = ops.WnDesc(cout, c1, c2, k, k, 1, 0)
        ops._call("vunet_weightnorm_bwd", ctypes.byref(wn), ops._p(slabs), ops._p(dshift), ns, ops._p(v), None, None,
                  None, None, ops._p(dw), None, ops._p(db), None, None, ops._p(work), 0, ops._stream())
        torch.cuda.synchronize()
        return dw, db, buf.value.decode()

    dwh, dbh, name = run(2)
    dw3, db3, name3 = run(1)
    assert name.startswith("conv_wgrad_direct_kernel"), name
    assert not name3.startswith("conv_wgrad_direct_kernel"), name3
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
    wz = torch.zeros(cout, ctot, k, k, dtype=torch.float64, requires_grad=True)
    (F.conv2d(torch.cat(xs, dim=1), wz, stride=stride, padding=pad) * dy.double().cpu()).sum().backward()
    ref = wz.grad
    sc = float(ref.abs().max())
    eh = float((dwh.double().cpu() - ref).abs().max())
    e3 = float((dw3.double().cpu() - ref).abs().max())
    assert eh <= 3e-6 * sc, (eh, sc)
    assert eh <= 5.0 * e3 + 4e-7 * sc, (eh, e3, sc)
    refb = dy.double().cpu().sum(dim=(0, 2, 3))
    assert float((dbh.double().cpu() - refb).abs().max()) <= 1e-5 * max(float(refb.abs().max()), 1.0)


# (n, c1, c2, cout, hs, ws, in_act, drop): stride-2 layers above the batching bound (N Ho Wo > 16384) -- conv_wgrad_s2_kernel
WGRAD_S2 = [
    (2, 32, 0, 64, 256, 256, 0, 0.0),      # the 256 -> 128 Downsample (two co tiles, six waves)
    (4, 64, 0, 128, 128, 160, 1, 0.05),    # 128-channel output (four co tiles, two units per wave), ELU + dropout, non-square
    (2, 32, 32, 32, 192, 256, 1, 0.05),    # two sources (two channel tiles), one co tile (three waves)
    (9, 32, 0, 64, 96, 96, 0, 0.0),        # splits that end inside a row / an image
]


@pytest.mark.parametrize("case", WGRAD_S2)
def test_staged_stride2_weight_gradient_vs_fp64_and_the_direct_kernel(case):
    """csrc/conv_wgrad_direct.hip, conv_wgrad_s2_kernel: x and dy staged once per step through LDS (column-parity planes) instead
    of read as 64-byte pieces per lane.  Against the float64 reference and against the direct kernel on the same inputs
    (tuning knob wgrad_rowsplit = 3 keeps these layers there)."""
    ops = _ops()
    n, c1, c2, cout, hs, ws, in_act, drop = case
    k, stride, pad = 3, 2, 1
    ho, wo = (hs + 2 - 3) // 2 + 1, (ws + 2 - 3) // 2 + 1
    g_ = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x1 = torch.randn(n, c1, hs, ws, generator=g_).cuda()
    x2 = torch.randn(n, c2, hs, ws, generator=g_).cuda() if c2 else None
    dy = torch.randn(n, cout, ho, wo, generator=g_).cuda()
    seed = 0xFACE
    ctot, ktot = c1 + c2, 9 * (c1 + c2)

    def run(knob):
        ops.set_tuning("wgrad_rowsplit", knob)
        wd = ops.WgradDesc(N=n, C1=c1, C2=c2, Hs=hs, Ws=ws, Cout=cout, Ho=ho, Wo=wo, KH=k, KW=k, stride=stride, pad=pad,
                           in_act=in_act, in_slope=0.0, drop_p=drop, drop_seed=seed, nsplit=1, flags=2)
        buf = ctypes.create_string_buffer(96)
        ops._call("vunet_conv2d_wgrad_variant", ctypes.byref(wd), buf, 96)
        ns = ops._lib.lib().vunet_conv2d_wgrad_nsplit(ctypes.byref(wd))
        wd.nsplit = ns
        cp = ops._r32(cout)
        slabs = torch.full((ns * cp * ktot + ns * cp,), float("nan"), device="cuda")
        dshift = slabs[ns * cp * ktot:]
        ops._call("vunet_conv2d_wgrad", ctypes.byref(wd), ops._p(x1), ops._p(x2), ops._p(dy), ops._p(slabs), ops._p(dshift),
                  ops._p(ops.absmax_partials(x1, x2)), ops._p(ops.absmax_partials(dy)), ops._stream())
        dw, db = torch.empty(cout, ctot, k, k, device="cuda"), torch.empty(cout, device="cuda")
        v, work = torch.zeros(cout, ctot, k, k, device="cuda"), torch.empty(cout * (ktot + 1), device="cuda")
        wn = ops.WnDesc(cout, c1, c2, k, k, 1, 0)
        ops._call("vunet_weightnorm_bwd", ctypes.byref(wn), ops._p(slabs), ops._p(dshift), ns, ops._p(v), None, None, None, None,
                  ops._p(dw), None, ops._p(db), None, None, ops._p(work), 0, ops._stream())
        torch.cuda.synchronize()
        return dw, db, buf.value.decode()
    dws, dbs, name_s = run(0)
    dwd, dbd, name_d = run(3)
    assert name_s == "conv_wgrad_s2_kernel" and name_d.startswith("conv_wgrad_direct_kernel<2"), (name_s, name_d)
    xs = []
    for i, x in enumerate((x1, x2)):
        if x is None:
            continue
        t = x.double().cpu()
        if in_act == ops.ACT_ELU:
            t = F.elu(t)
        if drop > 0:
            sd = seed if i == 0 else (seed + ops.SEED2_OFFSET) & 0xFFFFFFFF
            t = t * dropout_keep_mask(tuple(x.shape), drop, sd).double() * float(torch.tensor(1.0 / (1.0 - drop), dtype=torch.float32))
        xs.append(t)
    wz = torch.zeros(cout, ctot, k, k, dtype=torch.float64, requires_grad=True)
    (F.conv2d(torch.cat(xs, dim=1), wz, stride=2, padding=1) * dy.double().cpu()).sum().backward()
    ref = wz.grad
    sc = float(ref.abs().max())
    es, ed = float((dws.double().cpu() - ref).abs().max()), float((dwd.double().cpu() - ref).abs().max())
    assert es <= 3e-6 * sc, (es, sc)
    assert es <= 3.0 * ed + 4e-7 * sc, (es, ed, sc)
    refb = dy.double().cpu().sum(dim=(0, 2, 3))
    assert float((dbs.double().cpu() - refb).abs().max()) <= 1e-5 * max(float(refb.abs().max()), 1.0)


def test_batched_weight_gradient_launch_equals_one_launch_per_layer():
    """vunet_conv2d_wgrad_multi: the layers of WGRAD the library calls batchable, all in one call (several kernel forms, more
    items of one form than a launch holds), write the same slabs bit for bit as vunet_conv2d_wgrad_a2 per layer; layers the
    call does not take are refused, not silently skipped."""
    ops = _ops()
    ops.set_tuning("wgrad_rowsplit", 0)
    lib = ops._lib.lib()
    items, singles, keep = [], [], []
    cases = [c for c in WGRAD] + [WGRAD[0], WGRAD[1], WGRAD[8]] * 5   # 27 items: the 3x3 / 4-wide form 6x, the 1x1 form 8x ...
    refused = 0
    for ci, case in enumerate(cases):
        n, c1, c2, cout, hs, ws, k, stride, in_act, drop = case
        pad = 1 if k == 3 else 0
        ho, wo = (hs + 2 * pad - k) // stride + 1, (ws + 2 * pad - k) // stride + 1
        g_ = torch.Generator().manual_seed(1000 + ci)
        x1 = torch.randn(n, c1, hs, ws, generator=g_).cuda()
        x2 = torch.randn(n, c2, hs, ws, generator=g_).cuda() if c2 else None
        dy = torch.randn(n, cout, ho, wo, generator=g_).cuda()
        wd = ops.WgradDesc(N=n, C1=c1, C2=c2, Hs=hs, Ws=ws, Cout=cout, Ho=ho, Wo=wo, KH=k, KW=k, stride=stride, pad=pad,
                           in_act=in_act, in_slope=0.0, drop_p=drop, drop_seed=0xBEEF + ci, nsplit=1, flags=2)
        wd.nsplit = lib.vunet_conv2d_wgrad_nsplit(ctypes.byref(wd))
        if lib.vunet_conv2d_wgrad_batchable(ctypes.byref(wd)) != 1:
            refused += 1
            one = (ops.WgradItem * 1)()
            one[0].d = wd
            assert lib.vunet_conv2d_wgrad_multi(one, 1, ops._stream()) == -3
            continue
        ktot, cp = k * k * (c1 + c2), ops._r32(cout)
        size = wd.nsplit * cp * ktot + wd.nsplit * cp
        a, b = (torch.full((size,), float("nan"), device="cuda") for _ in range(2))
        amx, amd = ops.absmax_partials(x1, x2), ops.absmax_partials(dy)
        ops._call("vunet_conv2d_wgrad_a2", ctypes.byref(wd), ops._p(x1), ops._p(x2), ops._p(dy), ops._p(a),
                  ops._p(a[wd.nsplit * cp * ktot:]), ops._p(amx), None, ops._p(amd), ops._stream())
        items.append((wd, x1, x2, dy, b, b[wd.nsplit * cp * ktot:], amx, amd))
        singles.append(a)
        keep.append((x1, x2, dy, amx, amd))
    assert len(items) >= 20
    # a layer of the row-tiled weight-gradient kernel (3x3 / stride 1 on a 32-wide map) and a long direct launch (1x1 at 256^2)
    for shape in ((2, 32, 32, 32, 32, 3, 1), (16, 32, 32, 256, 256, 1, 0)):
        n, c, cout, hs, ws, k, pad = shape
        wd = ops.WgradDesc(N=n, C1=c, C2=0, Hs=hs, Ws=ws, Cout=cout, Ho=hs, Wo=ws, KH=k, KW=k, stride=1, pad=pad, in_act=0,
                           in_slope=0.0, drop_p=0.0, drop_seed=0, nsplit=1, flags=2)
        wd.nsplit = lib.vunet_conv2d_wgrad_nsplit(ctypes.byref(wd))
        assert lib.vunet_conv2d_wgrad_batchable(ctypes.byref(wd)) == 0
        one = (ops.WgradItem * 1)()
        one[0].d = wd
        assert lib.vunet_conv2d_wgrad_multi(one, 1, ops._stream()) == -3
    arr = (ops.WgradItem * len(items))()
    for i, (wd, x1, x2, dy, b, bsh, amx, amd) in enumerate(items):
        arr[i].d = wd
        arr[i].x1, arr[i].x2, arr[i].dy = x1.data_ptr(), None if x2 is None else x2.data_ptr(), dy.data_ptr()
        arr[i].slabs, arr[i].dshift = b.data_ptr(), bsh.data_ptr()
        arr[i].amax_x, arr[i].amax_x2, arr[i].amax_dy = amx.data_ptr(), None, amd.data_ptr()
    ops._call("vunet_conv2d_wgrad_multi", arr, len(items), ops._stream())
    torch.cuda.synchronize()
    for i, (a, it) in enumerate(zip(singles, items)):
        assert torch.equal(a.view(torch.int32), it[4].view(torch.int32)), f"item {i}: {cases[i]}"
