"""MI355X-native counterparts of the reference's ``lib/modules.py`` (lines 11-233).

Same class names, constructor signatures, state-dict keys and forward semantics as the reference
(citations are ``lib/modules.py:<line>`` of CompVis/behavior-driven-video-synthesis), but every
forward/backward is a hand-written gfx950 kernel reached through the C-ABI library
(``include/vunet_hip.h``).  What the reference issues as separate ATen ops -- weight-norm, conv,
``gamma*y+beta``, ELU, dropout, ``torch.cat``, the residual add, DepthToSpace -- is one fused
implicit-GEMM launch here (``ops.fused_conv``).
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Union

import torch
from torch import nn

from .. import ops

TensorOrPair = Union[torch.Tensor, Sequence[torch.Tensor]]


def _act_code(act_fn) -> tuple:
    """nn activation module -> (kernel activation code, slope)."""
    if act_fn is None or isinstance(act_fn, IDAct):
        return ops.ACT_NONE, 0.0
    if isinstance(act_fn, nn.ELU):
        if act_fn.alpha != 1.0:
            raise NotImplementedError("only ELU(alpha=1) is implemented in the HIP prologue")
        return ops.ACT_ELU, 0.0
    if isinstance(act_fn, nn.ReLU):
        return ops.ACT_RELU, 0.0
    if isinstance(act_fn, nn.LeakyReLU):
        return ops.ACT_LRELU, float(act_fn.negative_slope)
    raise NotImplementedError(f"activation {type(act_fn).__name__} has no HIP prologue")


def _split(x: TensorOrPair):
    if isinstance(x, (tuple, list)):
        assert len(x) == 2
        return x[0], x[1]
    return x, None


class SpaceToDepth(nn.Module):
    """lib/modules.py:11-21 (block-major channel order)."""

    def __init__(self, block_size):
        super().__init__()
        assert block_size == 2
        self.bs = block_size

    def forward(self, x):
        return ops.SpaceToDepth.apply(x)


class DepthToSpace(nn.Module):
    """lib/modules.py:24-34 (block-major channel order, not pixel_shuffle)."""

    def __init__(self, block_size):
        super().__init__()
        assert block_size == 2
        self.bs = block_size

    def forward(self, x):
        return ops.DepthToSpace.apply(x)


class UpsampleBilinear2x(nn.Module):
    """``nn.Upsample(scale_factor=2, mode="bilinear")`` as used at lib/modules.py:175 (parameter-free)."""

    def forward(self, x):
        return ops.UpsampleBilinear2x.apply(x)


class IDAct(nn.Module):
    """lib/modules.py:37-39."""

    def forward(self, input):
        return input


class _WeightNormConvParams(nn.Module):
    """Parameter holder with the key names of ``weight_norm(nn.Conv2d)``: bias, weight_g, weight_v."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = (kernel_size, kernel_size), (stride, stride), (padding, padding)
        w = torch.empty(out_channels, in_channels, kernel_size, kernel_size)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))  # nn.Conv2d.reset_parameters
        bound = 1.0 / math.sqrt(in_channels * kernel_size * kernel_size)
        self.bias = nn.Parameter(torch.empty(out_channels).uniform_(-bound, bound))
        self.weight_g = nn.Parameter(w.flatten(1).norm(dim=1).view(-1, 1, 1, 1).clone())  # g = ||v|| at init
        self.weight_v = nn.Parameter(w)


class NormConv2d(nn.Module):
    """lib/modules.py:120-145: ``gamma * conv(x; g v/||v||, b) + beta``.

    ``forward(x)`` is the reference call.  ``fused`` exposes the prologue/epilogue hooks the
    surrounding modules use (second source instead of torch.cat, pre-activation, dropout, residual,
    sub-pixel store).
    """

    kind = 0

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__()
        self.beta = nn.Parameter(torch.zeros([1, out_channels, 1, 1], dtype=torch.float32))
        self.gamma = nn.Parameter(torch.ones([1, out_channels, 1, 1], dtype=torch.float32))
        self.conv = _WeightNormConvParams(in_channels, out_channels, kernel_size, stride, padding)
        self.k, self.stride, self.padding = kernel_size, stride, padding

    def _params(self):
        return self.conv.weight_v, self.conv.weight_g, self.conv.bias, self.gamma, self.beta

    def fused(self, x: TensorOrPair, res: Optional[torch.Tensor] = None, in_act: int = ops.ACT_NONE,
              in_slope: float = 0.0, drop_p: float = 0.0, out_act: int = ops.ACT_NONE, d2s: bool = False,
              passthrough: bool = False):
        """``passthrough``: returns (y, x_alias) -- see ``ops.ConvCfg.passthrough``."""
        x1, x2 = _split(x)
        cfg = ops.ConvCfg(kind=self.kind, k=self.k, stride=self.stride, pad=self.padding, in_act=in_act,
                          in_slope=in_slope, drop_p=drop_p, drop_seed=ops.next_dropout_seed() if drop_p > 0 else 0,
                          out_act=out_act, d2s=d2s, owner=self, passthrough=passthrough)
        v, g, b, gamma, beta = self._params()
        return ops.fused_conv(x1, x2, res, v, g, b, gamma, beta, cfg)

    def forward(self, x):
        return self.fused(x)


class L2NormConv2d(nn.Module):
    """lib/modules.py:42-101 (``conv_layer_type: l2``), including the data-dependent init (:95-99)."""

    kind = 2

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, init=None):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.k = kernel_size
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size).normal_(0.0, 0.05))
        if bias:
            bound = 1 / math.sqrt(in_channels * kernel_size * kernel_size)
            self.bias = nn.Parameter(torch.empty(out_channels).uniform_(-bound, bound))
        else:
            self.bias = None
        self.beta = nn.Parameter(torch.zeros([1, out_channels, 1, 1], dtype=torch.float32))
        self.gamma = nn.Parameter(torch.ones([1, out_channels, 1, 1], dtype=torch.float32))
        self.init_fn = init if callable(init) else (lambda: False)

    def _params(self):
        return self.weight, None, self.bias, self.gamma, self.beta

    def fused(self, x: TensorOrPair, res=None, in_act=ops.ACT_NONE, in_slope=0.0, drop_p=0.0, out_act=ops.ACT_NONE,
              d2s=False):
        x1, x2 = _split(x)
        if self.init_fn() and self.training:
            # lib/modules.py:95-99: set gamma/beta from the batch statistics of the normalised conv output
            with torch.no_grad():
                cfg0 = ops.ConvCfg(kind=2, k=self.k, stride=self.stride, pad=self.padding, in_act=in_act,
                                   in_slope=in_slope)
                ones, zeros = torch.ones_like(self.gamma), torch.zeros_like(self.beta)
                y0 = ops.fused_conv(x1, x2, None, self.weight, None, self.bias, ones, zeros, cfg0)
                mean = y0.mean(dim=[0, 2, 3], keepdim=True)
                var = y0.var(dim=[0, 2, 3], keepdim=True)
                # in place: gamma / beta may be views of a flat optimiser bucket (optim.FlatBucket) -- rebinding .data
                # would detach them from the buffer Adam updates and from the prepacked pointer table
                self.gamma.data.copy_(1.0 / torch.sqrt(var + 1e-10))
                self.beta.data.copy_(-mean * self.gamma)
                ops.invalidate_active_prepack(self)
        cfg = ops.ConvCfg(kind=2, k=self.k, stride=self.stride, pad=self.padding, in_act=in_act, in_slope=in_slope,
                          drop_p=drop_p, drop_seed=ops.next_dropout_seed() if drop_p > 0 else 0, out_act=out_act,
                          d2s=d2s, owner=self)
        v, g, b, gamma, beta = self._params()
        return ops.fused_conv(x1, x2, res, v, g, b, gamma, beta, cfg)

    def forward(self, x, it=None):
        return self.fused(x)


class _PlainConvParams(nn.Module):
    """Parameter holder with nn.Conv2d's key names (weight, bias)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.k, self.stride, self.padding = kernel_size, stride, padding
        w = torch.empty(out_channels, in_channels, kernel_size, kernel_size)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        self.weight = nn.Parameter(w)
        if bias:
            bound = 1.0 / math.sqrt(in_channels * kernel_size * kernel_size)
            self.bias = nn.Parameter(torch.empty(out_channels).uniform_(-bound, bound))
        else:
            self.register_parameter("bias", None)

    def fused(self, x: TensorOrPair, res=None, in_act=ops.ACT_NONE, in_slope=0.0, drop_p=0.0, out_act=ops.ACT_NONE,
              d2s=False):
        x1, x2 = _split(x)
        cfg = ops.ConvCfg(kind=1, k=self.k, stride=self.stride, pad=self.padding, in_act=in_act, in_slope=in_slope,
                          drop_p=drop_p, drop_seed=ops.next_dropout_seed() if drop_p > 0 else 0, out_act=out_act,
                          d2s=d2s)
        return ops.fused_conv(x1, x2, res, self.weight, None, self.bias, None, None, cfg)

    def forward(self, x):
        return self.fused(x)


class Conv2d(_PlainConvParams):
    """Drop-in for the plain ``nn.Conv2d`` layers of the path (PatchGAN, VGG19, Regressor)."""


class LayerNormConv2d(nn.Module):
    """lib/modules.py:104-117: plain conv followed by InstanceNorm2d (affine=False, eps=1e-5)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__()
        self.conv = _PlainConvParams(in_channels, out_channels, kernel_size, stride, padding)
        self.norm = InstanceNorm2d(out_channels)

    def fused(self, x: TensorOrPair, res=None, in_act=ops.ACT_NONE, in_slope=0.0, drop_p=0.0, out_act=ops.ACT_NONE,
              d2s=False):
        y = self.norm(self.conv.fused(x, None, in_act, in_slope, drop_p))
        if out_act != ops.ACT_NONE:
            y = ops.Activation.apply(y, out_act, in_slope)
        if d2s:
            y = ops.DepthToSpace.apply(y)
        if res is not None:
            y = y + res
        return y

    def forward(self, x):
        return self.fused(x)


class InstanceNorm2d(nn.Module):
    """nn.InstanceNorm2d(affine=False, track_running_stats=False) on the wavefront-shuffle kernel."""

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features, self.eps = num_features, eps

    def forward(self, x):
        return ops.InstanceNorm.apply(x, self.eps)


class Downsample(nn.Module):
    """lib/modules.py:148-161: 3x3 stride-2 conv."""

    def __init__(self, channels, out_channels=None, conv_layer=NormConv2d):
        super().__init__()
        self.down = conv_layer(channels, channels if out_channels is None else out_channels, kernel_size=3,
                               stride=2, padding=1)

    def forward(self, x, passthrough: bool = False):
        """``passthrough`` (weight-normalised layers): -> (y, x_alias); a skip connection that also reads ``x`` should read
        the alias -- its gradient is then added in this layer's data-gradient kernel (``ops.ConvCfg.passthrough``)."""
        if passthrough and isinstance(self.down, NormConv2d):
            return self.down.fused(x, passthrough=True)
        return (self.down(x), x) if passthrough else self.down(x)


class Upsample(nn.Module):
    """lib/modules.py:164-182.  Sub-pixel branch: the DepthToSpace is the conv's store addressing."""

    def __init__(self, in_channels, out_channels, subpixel=True, conv_layer=NormConv2d):
        super().__init__()
        self.subpixel = subpixel
        if subpixel:
            self.up = conv_layer(in_channels, 4 * out_channels, 3, padding=1)
            self.op2 = DepthToSpace(block_size=2)
        else:
            # lib/modules.py:172-175: 3x3 conv to out_channels, then bilinear 2x up-sampling (subpixel_upsampling: False;
            # no shipped config selects it)
            self.up = conv_layer(in_channels, out_channels, 3, padding=1)
            self.op2 = UpsampleBilinear2x()

    def forward(self, x):
        if self.subpixel:
            return self.up.fused(x, d2s=True)
        return self.op2(self.up(x))


class VunetRNB(nn.Module):
    """lib/modules.py:185-233.

    ``out = x + conv3x3(dropout(act(cat(x, nin(act(a))))))`` as two fused launches: the 1x1 ``nin``
    with the activation as prologue, and the 3x3 conv reading (x, nin-out) as two sources with
    activation+dropout as prologue and ``+ x`` as epilogue.  ``a`` may be a pair of tensors, which
    is read as their channel concatenation (models/vunets.py:583 builds that concat).
    """

    def __init__(self, channels, a_channels=None, residual=False, kernel_size=3, activate=True,
                 conv_layer=NormConv2d, act_fn=None, dropout_prob=0.0):
        super().__init__()
        self.residual = residual
        self.dout = nn.Dropout(p=dropout_prob)
        if self.residual:
            assert a_channels is not None
            self.nin = conv_layer(a_channels, channels, kernel_size=1)
        if activate:
            self.act_fn = nn.ELU() if act_fn is None else act_fn
        else:
            self.act_fn = IDAct()
        in_c = 2 * channels if self.residual else channels
        self.conv = conv_layer(in_channels=in_c, out_channels=channels, kernel_size=kernel_size,
                               padding=kernel_size // 2)

    def forward(self, x, a: Optional[TensorOrPair] = None, passthrough: bool = False):
        """``passthrough`` (weight-normalised convolutions): -> (out, x_alias).  Whoever else reads ``x`` (a skip connection)
        should read the alias: its gradient then reaches this block's data-gradient kernel and is added in its epilogue --
        beside ``dy``, which the residual ``+ x`` puts there already (``ops.ConvCfg.passthrough``, the second residual slot
        of ``vunet_conv2d_a2``) -- instead of by an autograd add over the tensor."""
        act, slope = _act_code(self.act_fn)
        p = self.dout.p if self.training else 0.0
        kw = {"passthrough": True} if (passthrough and isinstance(self.conv, NormConv2d)) else {}
        if a is not None:
            assert self.residual
            a = self.nin.fused(a, in_act=act, in_slope=slope)
            out = self.conv.fused((x, a), res=x, in_act=act, in_slope=slope, drop_p=p, **kw)
        else:
            out = self.conv.fused(x, res=x, in_act=act, in_slope=slope, drop_p=p, **kw)
        if passthrough and not kw:
            return out, x
        return out


# ------------------------------------------------------------------------------------------------
# building blocks of the behaviour path (flow + recurrent nets): lib/modules.py:236-331
# ------------------------------------------------------------------------------------------------
class Linear(nn.Module):
    """Parameter holder with ``nn.Linear``'s keys (weight [out, in], bias [out]) and default initialisation; the product
    runs in ``seq.MlpGroup`` (csrc/seq.hip), never through ATen."""

    def __init__(self, in_features: int, out_features: int):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        w = torch.empty(out_features, in_features)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))   # nn.Linear.reset_parameters
        bound = 1.0 / math.sqrt(in_features)
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(torch.empty(out_features).uniform_(-bound, bound))


class _Marker(nn.Module):
    """A parameter-free slot of ``BasicFullyConnectedNet.main`` (LeakyReLU / Tanh): keeps the reference's indices, so the
    Linear layers sit at main.0, main.2, ... in the state dict.  The activation itself is applied inside the kernels."""

    def __init__(self, name: str):
        super().__init__()
        self.name = name

    def extra_repr(self):
        return self.name


class BasicFullyConnectedNet(nn.Module):
    """lib/modules.py:236-257: Linear, LeakyReLU, depth x (Linear, LeakyReLU), Linear [, Tanh]."""

    def __init__(self, dim, depth, hidden_dim=256, use_tanh=False, use_bn=False, out_dim=None):
        super().__init__()
        if use_bn:
            raise NotImplementedError("use_bn: no flow of the reference builds its MLPs with BatchNorm1d (models/flow/blocks.py)")
        self.use_tanh = use_tanh
        layers = [Linear(dim, hidden_dim), _Marker("LeakyReLU(0.01)")]
        for _ in range(depth):
            layers += [Linear(hidden_dim, hidden_dim), _Marker("LeakyReLU(0.01)")]
        layers.append(Linear(hidden_dim, dim if out_dim is None else out_dim))
        if use_tanh:
            layers.append(_Marker("Tanh"))
        self.main = nn.Sequential(*layers)
        self._engine = None

    def linears(self):
        return [m for m in self.main if isinstance(m, Linear)]

    def forward(self, x):
        """Stand-alone evaluation ([B, dim] -> [B, out]); inside a coupling block the s and t nets run as one launch per
        layer instead (``seq.FlowEngine``)."""
        from .. import seq
        if self._engine is None:
            object.__setattr__(self, "_engine", seq.MlpEngine(self))
        return self._engine(x)


class ActNorm(nn.Module):
    """lib/modules.py:260-331 for flat inputs ([B, C] or [B, C, 1, 1]): h = scale (x + loc), with the data-dependent
    initialisation on the first forward call (loc = -mean, scale = 1 / (std + 1e-6) over the batch)."""

    def __init__(self, num_features, logdet=False, affine=True):
        assert affine
        super().__init__()
        self.logdet = logdet
        self.loc = nn.Parameter(torch.zeros(1, num_features, 1, 1))
        self.scale = nn.Parameter(torch.ones(1, num_features, 1, 1))
        self.register_buffer("initialized", torch.tensor(0, dtype=torch.uint8))

    def initialize(self, input):
        from .. import seq
        seq.actnorm_initialize(self, input)

    def forward(self, input, reverse=False):
        from .. import seq
        if reverse:
            return self.reverse(input)
        if self.initialized.item() == 0:
            self.initialize(input)
            self.initialized.fill_(1)
        h, logdet = seq.actnorm_apply(self, input, False)
        return (h, logdet) if self.logdet else h

    def reverse(self, output):
        from .. import seq
        return seq.actnorm_apply(self, output, True)[0]
