"""MI355X-native counterparts of ``models/synth_discriminator.py`` (:10-256): PatchGANDiscriminator,
PartDiscriminator, DiscTrainer and compute_grad2, with the reference's constructor signatures and
state-dict keys.  Convolutions (4x4 / 3x3-valid / stride 2), InstanceNorm and LeakyReLU run in the
HIP kernels; LeakyReLU(0.2) is fused as the *next* convolution's prologue, which is the same function
as the reference's in-place activation on the producer side.

Upstream never instantiates any of this; parity is pinned per module
(tests/golden/g4_discriminators.npz).  The R1 penalty (``grad_pen=True``) needs a double backward
through the conv kernels and is not built yet: it raises.
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
    """Parameter-free placeholder that keeps nn.Sequential's indices (LeakyReLU slots)."""

    def __init__(self, kind):
        super().__init__()
        self.kind = kind


class PatchGANDiscriminator(nn.Module):
    """models/synth_discriminator.py:10-74 (pix2pix 70x70 PatchGAN)."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=nn.InstanceNorm2d):
        super().__init__()
        if type(norm_layer) == functools.partial:
            use_bias = norm_layer.func == nn.InstanceNorm2d
            norm_cls = norm_layer.func
        else:
            use_bias = norm_layer == nn.InstanceNorm2d
            norm_cls = norm_layer
        if norm_cls not in (nn.InstanceNorm2d, InstanceNorm2d):
            raise NotImplementedError("only InstanceNorm2d is implemented for the PatchGAN discriminator")
        kw, padw = 4, 1
        seq = [Conv2d(input_nc, ndf, kernel_size=kw, stride=2, padding=padw), _Slot("lrelu")]
        nf_mult = 1
        for n in range(1, n_layers):
            nf_prev, nf_mult = nf_mult, min(2 ** n, 8)
            seq += [Conv2d(ndf * nf_prev, ndf * nf_mult, kernel_size=kw, stride=2, padding=padw, bias=use_bias),
                    InstanceNorm2d(ndf * nf_mult), _Slot("lrelu")]
        nf_prev, nf_mult = nf_mult, min(2 ** n_layers, 8)
        seq += [Conv2d(ndf * nf_prev, ndf * nf_mult, kernel_size=kw, stride=1, padding=padw, bias=use_bias),
                InstanceNorm2d(ndf * nf_mult), _Slot("lrelu")]
        seq += [Conv2d(ndf * nf_mult, 1, kernel_size=kw, stride=1, padding=padw)]
        self.model = nn.Sequential(*seq)

    def forward(self, input):
        h, pending_lrelu = input, False
        for m in self.model:
            if isinstance(m, Conv2d):
                h = m.fused(h, in_act=ops.ACT_LRELU if pending_lrelu else ops.ACT_NONE, in_slope=0.2)
                pending_lrelu = False
            elif isinstance(m, InstanceNorm2d):
                h = m(h)
            else:
                pending_lrelu = True  # LeakyReLU(0.2, inplace): becomes the next conv's prologue
        return h


class PartDiscriminator(nn.Module):
    """models/synth_discriminator.py:77-112."""

    def __init__(self, n_scales, part_size, nf_in=3, conv_layer=NormConv2d, max_filters=256, dropout_prob=0.0):
        super().__init__()
        self.n_rnb = 2
        self.n_scales = n_scales
        self.nin = conv_layer(in_channels=nf_in, out_channels=16, kernel_size=3)
        blocks, nf, spatial_size = [], 16, part_size
        for _ in range(self.n_scales):
            blocks.append(VunetRNB(channels=nf, conv_layer=conv_layer, dropout_prob=dropout_prob))
            out_c = min(2 * nf, max_filters)
            blocks.append(Downsample(nf, out_c))
            nf = out_c
            spatial_size = spatial_size // 2
        self.feature_extractor = nn.Sequential(*blocks)
        self.n_linear_units = nf * spatial_size ** 2
        self.classifier = Linear(self.n_linear_units, 1)

    def forward(self, x):
        h = self.nin(x)
        h = self.feature_extractor(h)
        h = h.reshape(-1, self.n_linear_units)
        return self.classifier(h)


class DiscTrainer(object):
    """models/synth_discriminator.py:115-242."""

    def __init__(self, generator, config, discriminator=PartDiscriminator, grad_pen=False, lambda_gp=10,
                 grad_weighting=False, **kwargs):
        self.disc = discriminator(n_scales=config["pd_scales"], part_size=kwargs["spatial_size"] // 4)
        self.opt = None
        self.loss = nn.BCEWithLogitsLoss()
        if grad_pen:
            raise NotImplementedError("R1 gradient penalty needs double backward through the HIP conv kernels")
        self.use_gp = grad_pen
        self.lambda_gp = lambda_gp
        self.gw = grad_weighting
        self.adam_betas = config["adam_beta"]
        self.save_intervall = config["save_intervall"]
        self.generator = generator
        self.parallel = isinstance(generator, nn.DataParallel)

    def train_disc(self, real_x, fake_x, retain_graph=False):
        toggle_grad(self.disc, True)
        toggle_grad(self.generator, False)
        self.disc.train()
        self.opt.zero_grad()
        real_x.requires_grad_(True)
        disc_on_real = self.disc(real_x)
        real_loss = self.loss(disc_on_real, torch.ones_like(disc_on_real))
        real_loss.backward(retain_graph=retain_graph)
        fake_x.requires_grad_()
        disc_on_fake = self.disc(fake_x)
        fake_loss = self.loss(disc_on_fake, torch.zeros_like(disc_on_real))
        fake_loss.backward(retain_graph=True if self.gw else retain_graph)
        self.opt.step()
        toggle_grad(self.disc, False)
        toggle_grad(self.generator, True)
        dloss = real_loss + fake_loss
        return {"dloss": dloss.item(), "dloss_r": real_loss.item(), "dloss_f": fake_loss.item()}

    def get_genloss(self, x_fake, pre_loss, last_layer_weight):
        toggle_grad(self.generator, True)
        toggle_grad(self.disc, False)
        disc_on_fake = self.disc(x_fake)
        gen_loss = self.loss(disc_on_fake, torch.ones_like(disc_on_fake))
        if self.gw:
            self.generator.zero_grad()
            g_normal = torch.mean(autograd.grad(pre_loss, last_layer_weight, retain_graph=True)[0])
            self.generator.zero_grad()
            g_gen = torch.mean(autograd.grad(gen_loss, last_layer_weight, retain_graph=True)[0])
            loss_weight = torch.abs(g_normal / g_gen)
            loss_weight.requires_grad_(False)
        else:
            loss_weight = 1.0
        return gen_loss, loss_weight

    def init_training(self, devices, lr, d_ckpt=None, o_ckpt=None):
        if d_ckpt is not None:
            self.disc.load_state_dict(d_ckpt)
        self.disc.to(devices[0])
        self.opt = FusedAdam([{"params": list(self.disc.parameters()), "name": "disc"}], lr=lr,
                             betas=self.adam_betas)
        if o_ckpt is not None:
            self.opt.load_state_dict(o_ckpt)

    def update_lr(self, lr):
        for p in self.opt.param_groups:
            p["lr"] = lr

    def checkpoint(self, save_dir=None, trainer=None, name=None):
        """Save dict of :228-231 (the ignite handler wiring is out of scope)."""
        return {"disc": self.disc.state_dict(), "opt": self.opt.state_dict()}


def compute_grad2(d_out, x_in, allow_unused=False):
    """models/synth_discriminator.py:244-256 (R1 regulariser) -- requires double backward."""
    raise NotImplementedError("compute_grad2 needs double backward through the HIP conv kernels (SURVEY n2)")
