"""MI355X-native counterparts of ``models/synth_discriminator.py`` (:10-256): ``PatchGANDiscriminator``,
``PartDiscriminator``, ``DiscTrainer`` and ``compute_grad2`` with the reference's constructor signatures
and state-dict keys.  The 4x4 / valid-3x3 / stride-2 convolutions, InstanceNorm and LeakyReLU run in the HIP
kernels; ``LeakyReLU(0.2)`` is applied as the *next* convolution's prologue, which computes the same function
as the reference's in-place activation on the producer side.

Upstream never instantiates any of this, so parity is pinned per module
(tests/golden/g4_discriminators.npz), including the R1 penalty (``grad_pen=True``), whose double backward runs
through ``ops.ConvPlainFwd/Dgrad/Wgrad`` -- conv primitives that differentiate into each other.
"""
from __future__ import annotations

import functools

import torch
from torch import autograd, nn

from .. import ops
from ..lib.modules import Conv2d, Downsample, InstanceNorm2d, NormConv2d, VunetRNB
from ..lib.utils import toggle_grad
from ..optim import FusedAdam
from .vunets import Linear


class _Slot(nn.Module):
    """Parameter-free placeholder that keeps nn.Sequential's indices (the LeakyReLU positions)."""

    def __init__(self, kind):
        super().__init__()
        self.kind = kind


class PatchGANDiscriminator(nn.Module):
    """pix2pix 70x70 PatchGAN (:10-74): C64(s2) - [C128, C256](s2, IN) - C512(s1, IN) - C1(s1), all 4x4 pad 1."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=nn.InstanceNorm2d):
        super().__init__()
        norm_cls = norm_layer.func if isinstance(norm_layer, functools.partial) else norm_layer
        if norm_cls not in (nn.InstanceNorm2d, InstanceNorm2d):
            raise NotImplementedError("only InstanceNorm2d is implemented for the PatchGAN discriminator")
        biased = True  # InstanceNorm has no affine parameters, so the convs keep their bias (:22-27)
        # (in_mult, out_mult, stride, normalised) per conv; channel multipliers saturate at 8
        spec = [(None, 1, 2, False)]
        spec += [(min(2 ** (n - 1), 8), min(2 ** n, 8), 2, True) for n in range(1, n_layers)]
        spec += [(min(2 ** (n_layers - 1), 8), min(2 ** n_layers, 8), 1, True)]
        layers = []
        for in_mult, out_mult, stride, normed in spec:
            cin = input_nc if in_mult is None else ndf * in_mult
            layers.append(Conv2d(cin, ndf * out_mult, kernel_size=4, stride=stride, padding=1, bias=biased))
            if normed:
                layers.append(InstanceNorm2d(ndf * out_mult))
            layers.append(_Slot("lrelu"))
        layers.append(Conv2d(ndf * spec[-1][1], 1, kernel_size=4, stride=1, padding=1))
        self.model = nn.Sequential(*layers)

    def forward(self, input):
        h, lrelu_pending = input, False
        for layer in self.model:
            if isinstance(layer, _Slot):
                lrelu_pending = True        # becomes the next conv's prologue
            elif isinstance(layer, InstanceNorm2d):
                h = layer(h)
            else:
                h = layer.fused(h, in_act=ops.ACT_LRELU if lrelu_pending else ops.ACT_NONE, in_slope=0.2)
                lrelu_pending = False
        return h


class PartDiscriminator(nn.Module):
    """:77-112 -- valid 3x3 NormConv 3->16, then n_scales x {VunetRNB, stride-2 NormConv}, then one logit."""

    def __init__(self, n_scales, part_size, nf_in=3, conv_layer=NormConv2d, max_filters=256, dropout_prob=0.0):
        super().__init__()
        self.n_rnb, self.n_scales = 2, n_scales
        self.nin = conv_layer(in_channels=nf_in, out_channels=16, kernel_size=3)
        stages, width, side = [], 16, part_size
        for _ in range(n_scales):
            wider = min(2 * width, max_filters)
            stages += [VunetRNB(channels=width, conv_layer=conv_layer, dropout_prob=dropout_prob),
                       Downsample(width, wider)]
            width, side = wider, side // 2
        self.feature_extractor = nn.Sequential(*stages)
        self.n_linear_units = width * side * side
        self.classifier = Linear(self.n_linear_units, 1)

    def forward(self, x):
        feats = self.feature_extractor(self.nin(x))
        return self.classifier(feats.reshape(-1, self.n_linear_units))


class DiscTrainer(object):
    """:115-242 -- BCE-with-logits discriminator / generator steps around a ``PartDiscriminator``."""

    def __init__(self, generator, config, discriminator=PartDiscriminator, grad_pen=False, lambda_gp=10,
                 grad_weighting=False, **kwargs):
        self.disc = discriminator(n_scales=config["pd_scales"], part_size=kwargs["spatial_size"] // 4)
        self.generator = generator
        self.parallel = isinstance(generator, nn.DataParallel)
        self.loss = nn.BCEWithLogitsLoss()   # on [N, 1] logits: scalar bookkeeping, not a kernel
        self.use_gp, self.lambda_gp, self.gw = grad_pen, lambda_gp, grad_weighting
        self.adam_betas, self.save_intervall = config["adam_beta"], config["save_intervall"]
        self.opt = None
        self.averager = None                 # data parallel: set by init_training when a process group is up
        self.keep_generator_grads = False    # a caller that zeroes the generator's flat gradient buckets itself

    def _only(self, train_disc: bool):
        toggle_grad(self.disc, train_disc)
        toggle_grad(self.generator, not train_disc)

    def train_disc(self, real_x, fake_x, retain_graph=False, as_tensors=False):
        """``as_tensors``: return the losses as device scalars instead of python floats (the reference's ``.item()`` calls
        are three host synchronisations per step, and cannot be part of a captured hipGraph)."""
        self._only(True)
        self.disc.train()
        if self.averager is not None:
            self.averager.start_step()
        self.opt.zero_grad()
        logits_real = self.disc(real_x.requires_grad_(True))
        real_loss = self.loss(logits_real, torch.ones_like(logits_real))
        reg = None
        if self.use_gp:   # R1 penalty on the real batch (:148-151): double backward through the conv kernels
            real_loss.backward(create_graph=True)
            reg = self.lambda_gp * compute_grad2(logits_real, real_x).mean()
            reg.backward()
        else:
            real_loss.backward(retain_graph=retain_graph)
        logits_fake = self.disc(fake_x.requires_grad_())
        fake_loss = self.loss(logits_fake, torch.zeros_like(logits_real))
        fake_loss.backward(retain_graph=retain_graph or self.gw)
        if self.averager is not None:
            self.averager.finish()
        self.opt.step()
        self._only(False)
        out = {"dloss": (real_loss + fake_loss).detach(), "dloss_r": real_loss.detach(), "dloss_f": fake_loss.detach()}
        if reg is not None:
            out["gp"] = reg.detach()
        return out if as_tensors else {k: v.item() for k, v in out.items()}

    def get_genloss(self, x_fake, pre_loss, last_layer_weight):
        self._only(False)
        logits = self.disc(x_fake)
        gen_loss = self.loss(logits, torch.ones_like(logits))
        if not self.gw:
            return gen_loss, 1.0

        def mean_grad(loss):   # :197-205: mean gradient of a loss term at the generator's last layer
            if not self.keep_generator_grads:
                self.generator.zero_grad()
            with ops.direct_grads(False):   # the gradient must come back as a tensor, not land in .grad
                return torch.mean(autograd.grad(loss, last_layer_weight, retain_graph=True)[0])
        weight = torch.abs(mean_grad(pre_loss) / mean_grad(gen_loss))
        return gen_loss, weight.requires_grad_(False)

    def init_training(self, devices, lr, d_ckpt=None, o_ckpt=None, process_group=None):
        if d_ckpt is not None:
            self.disc.load_state_dict(d_ckpt)
        self.disc.to(devices[0])
        self.opt = FusedAdam([{"params": list(self.disc.parameters()), "name": "disc"}], lr=lr, betas=self.adam_betas)
        if o_ckpt is not None:
            self.opt.load_state_dict(o_ckpt)
        if torch.distributed.is_initialized():   # replaces nn.DataParallel(disc): replica + averaged gradients
            from ..parallel import BucketedGradAverager, broadcast_parameters
            broadcast_parameters(self.opt.buckets, 0, process_group)
            self.averager = BucketedGradAverager(self.opt.buckets, process_group)

    def update_lr(self, lr):
        for group in self.opt.param_groups:
            group["lr"] = lr

    def checkpoint(self, save_dir=None, trainer=None, name=None):
        """The save dict of :228-231 (the ignite handler wiring is out of scope)."""
        return {"disc": self.disc.state_dict(), "opt": self.opt.state_dict()}


def compute_grad2(d_out, x_in, allow_unused=False):
    """:244-256 -- R1 regulariser: per-sample squared norm of d(sum of logits)/d(input), differentiable."""
    (grad_dout,) = autograd.grad(outputs=d_out.sum(), inputs=x_in, create_graph=True, only_inputs=True)
    assert grad_dout.size() == x_in.size()
    return grad_dout.pow(2).reshape(x_in.size(0), -1).sum(1)
