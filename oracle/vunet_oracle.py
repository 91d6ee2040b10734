"""CPU oracle for the VUnet shape-and-posture hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import this
module: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the *checker*.

It is a functional (state-dict driven) restatement in plain PyTorch-CPU fp32
of the reference algorithm.  Every function cites the reference lines it
follows (paths relative to the upstream repository root).  Parity pin: the
restatement is checked against golden vectors that were produced by importing
the reference's own ``lib/modules.py`` / ``models/vunets.py`` / ``lib/losses.py``
/ ``models/synth_discriminator.py`` in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``,
``tests/test_oracle_golden.py``).

Parity unpinned (stated): the *pretrained* torchvision VGG19
weights (torchvision is absent; the topology is restated and pinned with
seeded synthetic weights).

All functions take a flat ``sd`` mapping with the reference's state-dict key
names (``eu.blocks.0.conv.conv.weight_v`` ...), so the product modules'
``state_dict()`` can be fed in unchanged.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# --------------------------------------------------------------------------
# index shuffles -- lib/modules.py:11-34
# --------------------------------------------------------------------------
def space_to_depth(x: Tensor, bs: int = 2) -> Tensor:
    """out[n,(i*bs+j)*C+c,h,w] = x[n,c,h*bs+i,w*bs+j]   (lib/modules.py:16-21)."""
    n, c, h, w = x.shape
    out = x.new_empty(n, bs * bs * c, h // bs, w // bs)
    for i in range(bs):
        for j in range(bs):
            out[:, (i * bs + j) * c:(i * bs + j + 1) * c] = x[:, :, i::bs, j::bs]
    return out


def depth_to_space(x: Tensor, bs: int = 2) -> Tensor:
    """out[n,c,h*bs+i,w*bs+j] = x[n,(i*bs+j)*C+c,h,w]   (lib/modules.py:29-34).

    Block-major channel order, i.e. NOT torch.pixel_shuffle.
    """
    n, c4, h, w = x.shape
    c = c4 // (bs * bs)
    parts = []
    for i in range(bs):
        row = []
        for j in range(bs):
            row.append(x[:, (i * bs + j) * c:(i * bs + j + 1) * c])
        # interleave along width
        parts.append(torch.stack(row, dim=-1).reshape(n, c, h, w * bs))
    # interleave along height
    return torch.stack(parts, dim=-2).reshape(n, c, h * bs, w * bs)


# --------------------------------------------------------------------------
# conv layers -- lib/modules.py:42-145
# --------------------------------------------------------------------------
def weight_norm_weight(v: Tensor, g: Tensor) -> Tensor:
    """w = g * v / ||v||, norm per output channel over (Cin,kh,kw).

    torch.nn.utils.weight_norm(dim=0) as used at lib/modules.py:135-138.
    """
    nrm = v.flatten(1).norm(dim=1).view(-1, 1, 1, 1)
    return v * (g / nrm)


def norm_conv(sd: SD, p: str, x: Tensor, stride: int = 1, padding: int = 0) -> Tensor:
    """NormConv2d.forward (lib/modules.py:140-145)."""
    w = weight_norm_weight(sd[p + ".conv.weight_v"], sd[p + ".conv.weight_g"])
    y = F.conv2d(x, w, sd[p + ".conv.bias"], stride=stride, padding=padding)
    return sd[p + ".gamma"] * y + sd[p + ".beta"]


def l2norm_conv(sd: SD, p: str, x: Tensor, stride: int = 1, padding: int = 0) -> Tensor:
    """L2NormConv2d.forward without the data-dependent init (lib/modules.py:89-101)."""
    w = sd[p + ".weight"]
    wn = w / w.flatten(1).norm(dim=1).clamp_min(1e-12).view(-1, 1, 1, 1)
    y = F.conv2d(x, wn, sd.get(p + ".bias"), stride=stride, padding=padding)
    return sd[p + ".gamma"] * y + sd[p + ".beta"]


def l2norm_conv_init(sd: SD, p: str, x: Tensor, stride: int = 1, padding: int = 0):
    """The data-dependent initialisation of L2NormConv2d (lib/modules.py:95-99, taken while ``init_fn()`` is true and the
    module trains): gamma = 1 / sqrt(var + 1e-10), beta = -mean * gamma from the statistics of the normalised convolution
    over (N, H, W) -- ``torch.var``: unbiased -- then the layer's output with them.  -> (y, gamma, beta)."""
    w = sd[p + ".weight"]
    wn = w / w.flatten(1).norm(dim=1).clamp_min(1e-12).view(-1, 1, 1, 1)
    y = F.conv2d(x, wn, sd.get(p + ".bias"), stride=stride, padding=padding)
    mean = y.mean(dim=[0, 2, 3], keepdim=True)
    var = y.var(dim=[0, 2, 3], keepdim=True)
    gamma = 1.0 / torch.sqrt(var + 1e-10)
    beta = -mean * gamma
    return gamma * y + beta, gamma, beta


def instance_norm(x: Tensor, eps: float = 1e-5) -> Tensor:
    """nn.InstanceNorm2d(affine=False, track_running_stats=False)."""
    m = x.mean(dim=(2, 3), keepdim=True)
    v = x.var(dim=(2, 3), unbiased=False, keepdim=True)
    return (x - m) / torch.sqrt(v + eps)


def layernorm_conv(sd: SD, p: str, x: Tensor, stride: int = 1, padding: int = 0) -> Tensor:
    """LayerNormConv2d.forward (lib/modules.py:114-117)."""
    y = F.conv2d(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"], stride=stride, padding=padding)
    return instance_norm(y)


def _conv_any(sd: SD, p: str, x: Tensor, stride: int = 1, padding: int = 0) -> Tensor:
    """Dispatch on the keys present (l1 / l2 / ln variants)."""
    if p + ".conv.weight_v" in sd:
        return norm_conv(sd, p, x, stride, padding)
    if p + ".weight" in sd:
        return l2norm_conv(sd, p, x, stride, padding)
    return layernorm_conv(sd, p, x, stride, padding)


def downsample(sd: SD, p: str, x: Tensor) -> Tensor:
    """Downsample.forward (lib/modules.py:160-161): 3x3 stride 2 pad 1."""
    return _conv_any(sd, p + ".down", x, stride=2, padding=1)


def upsample_bilinear(sd: SD, p: str, x: Tensor) -> Tensor:
    """Upsample.forward, non-sub-pixel branch (lib/modules.py:172-182): 3x3 NormConv then bilinear 2x (align_corners
    False, what nn.Upsample(mode="bilinear") does when the argument is left at its default)."""
    y = _conv_any(sd, p + ".up", x, padding=1)
    return F.interpolate(y, scale_factor=2, mode="bilinear", align_corners=False)


def upsample(sd: SD, p: str, x: Tensor) -> Tensor:
    """Upsample.forward, sub-pixel branch (lib/modules.py:169-171,179-182)."""
    return depth_to_space(_conv_any(sd, p + ".up", x, stride=1, padding=1), 2)


def dropout_apply(x: Tensor, mask: Optional[Tensor], p: float) -> Tensor:
    """nn.Dropout with an injected keep-mask (1 = keep)."""
    if mask is None or p == 0.0:
        return x
    return x * mask / (1.0 - p)


def rnb(sd: SD, p: str, x: Tensor, a: Optional[Tensor] = None,
        drop_mask: Optional[Tensor] = None, drop_p: float = 0.0, drop=None) -> Tensor:
    """VunetRNB.forward (lib/modules.py:221-233), ELU activation, 3x3 conv.

    ``drop`` (whole-model parity tests): ``drop(p, x_shape, a_shape) -> (keep-mask over cat(x, nin(a)), prob)`` for the
    block named ``p`` -- the keep-mask nn.Dropout would have drawn at :229, injected."""
    r = x
    if a is not None:
        a = _conv_any(sd, p + ".nin", F.elu(a))
        r = torch.cat([r, a], dim=1)
    if drop is not None:
        drop_mask, drop_p = drop(p, tuple(x.shape), None if a is None else tuple(a.shape))
    r = dropout_apply(F.elu(r), drop_mask, drop_p)
    r = _conv_any(sd, p + ".conv", r, padding=1)
    return x + r


# --------------------------------------------------------------------------
# architecture bookkeeping -- models/vunets.py:427-488
# --------------------------------------------------------------------------
def vunet_dims(cfg: dict, n_channels_x: int = 3) -> dict:
    ns = cfg.get("n_scales", 0)
    n_scales = (1 + int(round(math.log2(cfg["spatial_size"]))) - cfg["bottleneck_factor"]) if ns < 6 else ns
    n_scales_x = n_scales - cfg["box_factor"] if n_channels_x > 3 else n_scales
    return dict(n_scales=n_scales, n_scales_x=n_scales_x, nf_start=cfg["nf_start"], nf_max=cfg["nf_max"],
                n_latent_scales=cfg["n_latent_scales"])


def enc_up(sd: SD, p: str, x: Tensor, n_scales: int, drop=None) -> List[Tensor]:
    """EncUp.forward / DecUp.forward (models/vunets.py:133-148, 246-261)."""
    hs = []
    h = _conv_any(sd, p + ".nin", x)
    for i in range(n_scales):
        for n in range(2):
            h = rnb(sd, f"{p}.blocks.{2 * i + n}", h, drop=drop)
            hs.append(h)
        if i + 1 < n_scales:
            h = downsample(sd, f"{p}.downs.{i}", h)
    return hs


def enc_down_alter(sd: SD, p: str, gs: Sequence[Tensor], n_latent: int,
                   eps: Optional[Sequence[Tensor]] = None, drop=None):
    """EncDownAlter.forward (models/vunets.py:558-597).

    ``eps`` replaces torch.randn_like at :596 (one tensor per latent scale).
    Does not mutate ``gs`` (the reference pops from the caller's list).
    """
    gs = list(gs)
    hs, means, logstds, zs = [], [], [], []
    h = _conv_any(sd, p + ".nin", gs[-1])
    for i in range(n_latent):
        h = rnb(sd, f"{p}.blocks.{2 * i}", h, gs.pop(), drop=drop)
        hs.append(h)
        mu = _conv_any(sd, f"{p}.make_latent_params.{i}", h, padding=1)
        ls = torch.sigmoid(_conv_any(sd, f"{p}.make_logstds.{i}", h, padding=1))
        means.append(mu)
        logstds.append(ls)
        e = eps[i] if eps is not None else torch.randn_like(mu)
        z = e * torch.exp(ls) + mu
        zs.append(z)
        gz = torch.cat([gs.pop(), z], dim=1)
        h = rnb(sd, f"{p}.blocks.{2 * i + 1}", h, gz, drop=drop)
        hs.append(h)
        h = upsample(sd, f"{p}.ups.{i}", h)
    h = rnb(sd, p + ".fin_block", h, gs.pop(), drop=drop)
    hs.append(h)
    return hs, means, logstds, zs


def dec_down_alter(sd: SD, p: str, gs: Sequence[Tensor], zs: Sequence[Tensor], n_scales: int,
                   n_latent: int, training: bool = True,
                   prior_eps: Optional[Sequence[Tensor]] = None, subpixel: bool = True, drop=None) -> Tensor:
    """DecDownAlter.forward (models/vunets.py:332-414).  ``subpixel`` False (``subpixel_upsampling: False``): the levels
    past the latent scales up-sample bilinearly (models/vunets.py:325-329)."""
    gs = list(gs)
    zs = list(zs)
    h = _conv_any(sd, p + ".nin", gs[-1])
    for i in range(n_scales):
        h = rnb(sd, f"{p}.blocks.{2 * i}", h, gs.pop(), drop=drop)
        if i < n_latent:
            if training:
                z = zs.pop(0)
            else:
                z = prior_eps[i] if prior_eps is not None else torch.randn_like(h)
            h = rnb(sd, f"{p}.auto_blocks.{i}", h, z, drop=drop)
        h = rnb(sd, f"{p}.blocks.{2 * i + 1}", h, gs.pop(), drop=drop)
        if i + 1 < n_scales:
            h = upsample(sd, f"{p}.ups.{i}", h) if (subpixel or i < n_latent) else upsample_bilinear(sd, f"{p}.ups.{i}", h)
    assert not gs
    return _conv_any(sd, p + ".out_conv", h, padding=1)


def vunet_alter_forward(sd: SD, cfg: dict, x: Tensor, c: Tensor,
                        eps: Optional[Sequence[Tensor]] = None, n_channels_x: int = 3, drop=None):
    """VunetAlter.forward (models/vunets.py:490-500) -> (img, means, logstds, hs).  ``drop``: see ``rnb``."""
    d = vunet_dims(cfg, n_channels_x)
    hs = enc_up(sd, "eu", x, d["n_scales_x"], drop=drop)
    _, means, logstds, zs = enc_down_alter(sd, "ed", hs, d["n_latent_scales"], eps, drop=drop)
    gs = enc_up(sd, "du", c, d["n_scales"], drop=drop)
    img = dec_down_alter(sd, "dd", gs, zs, d["n_scales"], d["n_latent_scales"], True,
                         subpixel=bool(cfg.get("subpixel_upsampling", True)), drop=drop)
    return img, means, logstds, hs


def vunet_alter_transfer(sd: SD, cfg: dict, x: Tensor, c: Tensor,
                         eps: Optional[Sequence[Tensor]] = None, n_channels_x: int = 3) -> Tensor:
    """VunetAlter.transfer (models/vunets.py:508-515): posterior means as z."""
    d = vunet_dims(cfg, n_channels_x)
    hs = enc_up(sd, "eu", x, d["n_scales_x"])
    _, means, _, _ = enc_down_alter(sd, "ed", hs, d["n_latent_scales"], eps)
    gs = enc_up(sd, "du", c, d["n_scales"])
    return dec_down_alter(sd, "dd", gs, list(means), d["n_scales"], d["n_latent_scales"], True)


def vunet_alter_test_forward(sd: SD, cfg: dict, c: Tensor, prior_eps: Sequence[Tensor]) -> Tensor:
    """VunetAlter.test_forward (models/vunets.py:502-506)."""
    d = vunet_dims(cfg)
    gs = enc_up(sd, "du", c, d["n_scales"])
    return dec_down_alter(sd, "dd", gs, [], d["n_scales"], d["n_latent_scales"], False, prior_eps)


# --------------------------------------------------------------------------
# VunetOrg -- models/vunets.py:18-106, 159-219, 600-783
# --------------------------------------------------------------------------
def enc_down_org(sd: SD, p: str, gs: Sequence[Tensor], n_latent: int,
                 eps: Optional[Sequence[Tensor]] = None):
    """EncDown.forward (models/vunets.py:191-219); z = q + eps (:151-156)."""
    gs = list(gs)
    hs, qs, zs = [], [], []
    h = _conv_any(sd, p + ".nin", gs[-1])
    for i in range(n_latent):
        h = rnb(sd, f"{p}.blocks.{2 * i}", h, gs.pop())
        hs.append(h)
        q = _conv_any(sd, f"{p}.make_latent_params.{i}", h, padding=1)
        qs.append(q)
        z = q + (eps[i] if eps is not None else torch.randn_like(q))
        zs.append(z)
        h = rnb(sd, f"{p}.blocks.{2 * i + 1}", h, torch.cat([gs.pop(), z], dim=1))
        hs.append(h)
        h = upsample(sd, f"{p}.ups.{i}", h)
    h = rnb(sd, p + ".fin_block", h, gs.pop())
    hs.append(h)
    return hs, qs, zs


def dec_down_org(sd: SD, p: str, gs: Sequence[Tensor], zs_posterior: Sequence[Tensor], n_scales: int,
                 n_latent: int, training: bool = True,
                 prior_eps: Optional[Sequence[Sequence[Tensor]]] = None):
    """DecDown.forward (models/vunets.py:704-774): 4-group autoregressive prior.

    ``prior_eps[i][l]`` replaces randn_like in latent_sample for scale i, group l.
    """
    gs = list(gs)
    zs_posterior = list(zs_posterior)
    hs, ps, zs = [], [], []
    h = _conv_any(sd, p + ".nin", gs[-1])
    for i in range(n_scales):
        h = rnb(sd, f"{p}.blocks.{2 * i}", h, gs.pop())
        hs.append(h)
        if i < n_latent:
            sc = f"l_{i}"
            if training:
                zp = zs_posterior[0]
                groups = list(torch.split(space_to_depth(zp), zp.shape[1], dim=1))
            p_groups, z_groups = [], []
            pre = rnb(sd, f"{p}.auto_blocks.{sc}.0", h)
            pf = space_to_depth(pre)
            for l in range(4):
                pg = _conv_any(sd, f"{p}.auto_lp.{sc}.{l}", pf, padding=1)
                p_groups.append(pg)
                e = prior_eps[i][l] if prior_eps is not None else torch.randn_like(pg)
                zg = pg + e
                z_groups.append(zg)
                fb = groups.pop(0) if training else zg
                if l + 1 < 4:
                    pf = rnb(sd, f"{p}.auto_blocks.{sc}.{l + 1}", pf, fb)
            ps.append(depth_to_space(torch.cat(p_groups, dim=1)))
            z_prior = depth_to_space(torch.cat(z_groups, dim=1))
            zs.append(z_prior)
            z = zs_posterior.pop(0) if training else z_prior
            h = _conv_any(sd, f"{p}.latent_nins.{sc}", torch.cat([h, z], dim=1))
            h = rnb(sd, f"{p}.blocks.{2 * i + 1}", h, gs.pop())
            hs.append(h)
        else:
            h = rnb(sd, f"{p}.blocks.{2 * i + 1}", h, gs.pop())
            hs.append(h)
        if i + 1 < n_scales:
            h = upsample(sd, f"{p}.ups.{i}", h)
    assert not gs
    return _conv_any(sd, p + ".out_conv", hs[-1], padding=1), hs, ps, zs


def vunet_org_forward(sd: SD, cfg: dict, x: Tensor, c: Tensor, eps=None, prior_eps=None,
                      n_channels_x: int = 3):
    """VunetOrg.forward (models/vunets.py:81-91) -> (img, qs, ps)."""
    d = vunet_dims(cfg, n_channels_x)
    hs = enc_up(sd, "eu", x, d["n_scales_x"])
    _, qs, zs = enc_down_org(sd, "ed", hs, d["n_latent_scales"], eps)
    gs = enc_up(sd, "du", c, d["n_scales"])
    img, _, ps, _ = dec_down_org(sd, "dd", gs, zs, d["n_scales"], d["n_latent_scales"], True, prior_eps)
    return img, qs, ps


# --------------------------------------------------------------------------
# Regressor -- models/vunets.py:786-824
# --------------------------------------------------------------------------
def regressor(sd: SD, embeddings: Sequence[Tensor], n_linear: int = 2) -> Tensor:
    out = []
    for k, e in enumerate(reversed(list(embeddings))):
        y = F.relu(F.conv2d(e, sd[f"embedders.{k}.weight"], sd[f"embedders.{k}.bias"]))
        out.append(y.flatten(1))
    o = torch.cat(out, dim=-1)
    for i in range(n_linear):
        o = F.linear(o, sd[f"linears.{i}.weight"], sd[f"linears.{i}.bias"])
        if i < n_linear - 1:
            o = F.relu(o)
    return o


# --------------------------------------------------------------------------
# losses -- lib/losses.py:26-37, 55-119, 283-291
# --------------------------------------------------------------------------
def kl_loss(mu: Tensor, logstd: Tensor) -> Tensor:
    """lib/losses.py:283-291 (inputs already flattened to [N, D])."""
    d = mu.shape[1]
    kl = torch.sum(-logstd + 0.5 * (torch.exp(logstd) ** 2 + mu ** 2), dim=-1) - 0.5 * d
    return kl.mean()


def compute_kl_with_prior(means: Sequence[Tensor], logstds: Sequence[Tensor]) -> Tensor:
    """lib/losses.py:68-78: mean over latent scales of the per-scale batch-mean KL."""
    terms = [kl_loss(m.reshape(m.shape[0], -1), l.reshape(l.shape[0], -1)) for m, l in zip(means, logstds)]
    return torch.stack(terms).mean()


def latent_kl(prior_mean: Tensor, posterior_mean: Tensor) -> Tensor:
    """lib/losses.py:26-37."""
    return (0.5 * (prior_mean - posterior_mean) ** 2).sum(dim=(1, 2, 3)).mean()


def compute_kl_loss(prior_means: Sequence[Tensor], posterior_means: Sequence[Tensor]) -> Tensor:
    """lib/losses.py:55-65 (sum over scales)."""
    return torch.stack([latent_kl(p, q) for p, q in zip(prior_means, posterior_means)]).sum()


VGG19_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M",
             512, 512, 512, 512, "M"]
VGG_TAPS = {3: "relu1_2", 8: "relu2_2", 13: "relu3_2", 22: "relu4_2", 31: "relu5_2"}
VGG_MEAN = (0.485, 0.456, 0.406)
VGG_STD = (0.229, 0.224, 0.225)


def vgg19_feature_layout(cfg: Sequence = VGG19_CFG) -> List[Tuple[str, int, int]]:
    """torchvision vgg19 ``features`` module list (cfg 'E'): (kind, cin, cout) per index."""
    mods, cin = [], 3
    for v in cfg:
        if v == "M":
            mods.append(("pool", cin, cin))
        else:
            mods.append(("conv", cin, v))
            mods.append(("relu", v, v))
            cin = v
    return mods


def make_synthetic_vgg19(seed: int = 1234, width_div: int = 1) -> SD:
    """Seeded He-normal VGG19 ``features`` weights with torchvision key names.

    ``width_div`` shrinks the channel widths (test-size networks).
    """
    g = torch.Generator().manual_seed(seed)
    sd: SD = {}
    cfg = [v if v == "M" else max(v // width_div, 4) for v in VGG19_CFG]
    for idx, (kind, cin, cout) in enumerate(vgg19_feature_layout(cfg)):
        if kind == "conv":
            std = math.sqrt(2.0 / (cin * 9))
            sd[f"features.{idx}.weight"] = torch.randn(cout, cin, 3, 3, generator=g) * std
            sd[f"features.{idx}.bias"] = (torch.rand(cout, generator=g) - 0.5) * 0.1
    return sd


def perceptual_vgg(vgg_sd: SD, x: Tensor, last: int = 36) -> Dict[str, Tensor]:
    """PerceptualVGG.forward (models/imagenet_pretrained.py:42-61).

    Runs ``features`` 0..last; returns dict input, relu1_2 .. relu5_2 in that order.
    (The reference iterates all 37 modules; outputs past index 31 are discarded,
    so ``last=31`` gives identical results.)
    """
    mean = torch.tensor(VGG_MEAN, dtype=x.dtype).view(1, 3, 1, 1)
    std = torch.tensor(VGG_STD, dtype=x.dtype).view(1, 3, 1, 1)
    x = ((x + 1.0) / 2.0 - mean) / std
    out = {"input": x}
    n_conv = sum(1 for k in vgg_sd if k.endswith(".weight"))
    layout = vgg19_feature_layout()
    assert n_conv == 16
    for idx, (kind, _, _) in enumerate(layout):
        if idx > last:
            break
        if kind == "conv":
            x = F.conv2d(x, vgg_sd[f"features.{idx}.weight"], vgg_sd[f"features.{idx}.bias"], padding=1)
        elif kind == "relu":
            x = F.relu(x)
        else:
            x = F.max_pool2d(x, 2, 2)
        if idx in VGG_TAPS:
            out[VGG_TAPS[idx]] = x
    return out


def vgg_loss(vgg_sd: SD, loss_weights: Sequence[float], target: Tensor, pred: Tensor) -> Dict[str, Tensor]:
    """lib/losses.py:81-102: w_i * mean|t_i - p_i| per tap, each of shape [1]."""
    tf = perceptual_vgg(vgg_sd, target, last=31)
    pf = perceptual_vgg(vgg_sd, pred, last=31)
    return {k: (loss_weights[i] * (tf[k] - pf[k]).abs().mean()).unsqueeze(-1) for i, k in enumerate(pf)}


# --------------------------------------------------------------------------
# discriminators -- models/synth_discriminator.py:10-112, 244-256
# --------------------------------------------------------------------------
def part_discriminator(sd: SD, x: Tensor, n_scales: int) -> Tensor:
    """PartDiscriminator.forward (models/synth_discriminator.py:105-112)."""
    h = norm_conv(sd, "nin", x)  # 3x3, no padding (:85)
    for i in range(n_scales):
        h = rnb(sd, f"feature_extractor.{2 * i}", h)
        h = downsample(sd, f"feature_extractor.{2 * i + 1}", h)
    return F.linear(h.reshape(h.shape[0], -1), sd["classifier.weight"], sd["classifier.bias"])


def patchgan_discriminator(sd: SD, x: Tensor, n_layers: int = 3) -> Tensor:
    """PatchGANDiscriminator.forward with InstanceNorm (models/synth_discriminator.py:30-74)."""
    h = F.leaky_relu(F.conv2d(x, sd["model.0.weight"], sd["model.0.bias"], stride=2, padding=1), 0.2)
    idx = 2
    for n in range(1, n_layers + 1):
        stride = 2 if n < n_layers else 1
        h = F.conv2d(h, sd[f"model.{idx}.weight"], sd.get(f"model.{idx}.bias"), stride=stride, padding=1)
        h = F.leaky_relu(instance_norm(h), 0.2)
        idx += 3
    return F.conv2d(h, sd[f"model.{idx}.weight"], sd[f"model.{idx}.bias"], stride=1, padding=1)


def bce_with_logits(logits: Tensor, target_value: float) -> Tensor:
    """nn.BCEWithLogitsLoss against a constant target (models/synth_discriminator.py:128,147,161)."""
    t = torch.full_like(logits, target_value)
    return F.binary_cross_entropy_with_logits(logits, t)


# --------------------------------------------------------------------------
# training-step scalars -- experiments/shape_and_pose_net.py:82-85, 311-319, lib/utils.py:520-527
# --------------------------------------------------------------------------
def linear_var(act_it, start_it, end_it, start_val, end_val, clip_min, clip_max):
    v = float(end_val - start_val) / (end_it - start_it) * (act_it - start_it) + start_val
    return min(max(v, clip_min), clip_max)


def update_gamma(gamma: float, gamma_step: float, imax: float, kl: float) -> float:
    return max(gamma - gamma_step * (imax - kl), 0.0)


def train_step_losses(sd: SD, cfg: dict, vgg_sd: SD, vgg_weights, x: Tensor, c: Tensor, target: Tensor,
                      eps, gamma: float, iteration: int, n_init_batches: int, ll_weight: float = 1.0, drop=None):
    """Loss assembly of train_fn (experiments/shape_and_pose_net.py:382-405), regressor path off."""
    img, means, logstds, _ = vunet_alter_forward(sd, cfg, x, c, eps, drop=drop)
    ld = vgg_loss(vgg_sd, vgg_weights, target, img)
    ll = ll_weight * torch.stack(list(ld.values()), dim=0).sum()
    kl = compute_kl_with_prior(means, logstds)
    loss = ll
    if iteration > n_init_batches:
        loss = loss + gamma * kl
    return loss, ll, kl, img


def regressor_side_loop(sd: SD, cfg: dict, reg_sd: SD, opt_reg, reg_imgs: Tensor, reg_targets: Tensor,
                        reg_eps: Optional[Sequence[Sequence[Tensor]]] = None) -> Tuple[Tensor, List[float]]:
    """The regressor side loop of train_fn (experiments/shape_and_pose_net.py:407-425): for every i in
    ``reg_imgs.shape[1]`` the frozen encoder ``ed(eu(reg_imgs[:, i]))`` (no_grad, :413), the regressor on its means,
    the L2-norm loss (:417-419) and one step of the regressor's own Adam (:420-422).  Returns the LAST step's loss
    (the one the caller clamps at 1.2 and scales by ``weight_regressor``, :424-425) and every step's value."""
    d = vunet_dims(cfg)
    values, loss_regressor = [], None
    for i in range(reg_imgs.shape[1]):
        with torch.no_grad():
            hs = enc_up(sd, "eu", reg_imgs[:, i], d["n_scales_x"])
            _, means, _, _ = enc_down_alter(sd, "ed", hs, d["n_latent_scales"], None if reg_eps is None else reg_eps[i])
        preds = regressor(reg_sd, means)
        tgts = reg_targets[:, i].reshape(reg_targets.shape[0], -1)
        loss_regressor = torch.norm(preds - tgts, dim=1).mean()
        opt_reg.zero_grad()
        loss_regressor.backward()
        opt_reg.step()
        values.append(float(loss_regressor))
    return loss_regressor.detach(), values
