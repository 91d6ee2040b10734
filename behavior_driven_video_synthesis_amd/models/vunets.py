"""MI355X-native counterparts of the reference's ``models/vunets.py``.

Public surface kept for drop-in use by ``experiments.shape_and_pose_net`` / ``experiments.vunet``:
class names, constructor keywords, attributes (``eu ed du dd n_scales spatial_size n_channels_x``),
return tuples and -- because optimizer checkpoints are index-mapped -- the parameter registration
order, hence the state-dict keys *and their order* (checked against the reference's own key list in
tests/test_cabi_and_host.py).  ``models/vunets.py:<line>`` citations point at the upstream code each
piece reproduces.  All arithmetic runs in the fused gfx950 kernels behind ``lib/modules.py``.

Deliberate, caller-invisible differences: ``torch.cat`` of skip/latent tensors is never materialised
(the convolutions read two sources), the reparametrisation is one fused kernel, the bottleneck
encoders work on a copy of the caller's feature list instead of popping from it, and every sampling
site accepts an injected noise tensor (used by the parity tests).
"""
from __future__ import annotations

from functools import partial
from typing import Optional, Sequence

import numpy as np
import torch
from torch import nn

from .. import ops
from ..lib.modules import (Conv2d, DepthToSpace, Downsample, L2NormConv2d, LayerNormConv2d, NormConv2d, SpaceToDepth,
                           Upsample, VunetRNB)

RNB_PER_SCALE = 2


def _count_scales(cfg) -> int:
    """:22-30 / :430-438 -- explicit ``n_scales`` >= 6 wins, else 1 + log2(size) - bottleneck_factor."""
    if cfg["n_scales"] >= 6:
        return cfg["n_scales"]
    return 1 + int(np.round(np.log2(cfg["spatial_size"]))) - cfg["bottleneck_factor"]


def _pick_conv_layer(cfg, init_fn, layer_norm_ok: bool):
    """:37-44 / :445-453 -> (layer factory, name used in the start-up print)."""
    kind = cfg["conv_layer_type"]
    if kind == "l1":
        return NormConv2d, "L1NormConv2d"
    if kind == "l2":
        return partial(L2NormConv2d, init=init_fn, bias=False), "L2NormConv2d"
    if not layer_norm_ok:
        raise NotImplementedError("No conv layers others than l1 and l2 normalized ones are available.")
    return LayerNormConv2d, "LayerNormConv2d"


def _skip_block(width, a_width, conv_layer=NormConv2d, p=0.0):
    return VunetRNB(channels=width, a_channels=a_width, residual=True, conv_layer=conv_layer, dropout_prob=p)


def _noise_like(t, given):
    return torch.randn_like(t) if given is None else given


# ------------------------------------------------------------------------------------------------
# feature pyramids  (EncUp :109-148, DecUp :222-261 -- the same stack on two different inputs)
# ------------------------------------------------------------------------------------------------
class _Pyramid(nn.Module):
    def __init__(self, n_scales, n_filters, max_filters, nf_in=3, conv_layer=NormConv2d, dropout_prob=0.0):
        super().__init__()
        self.n_rnb, self.n_scales = RNB_PER_SCALE, n_scales
        self.nin = conv_layer(in_channels=nf_in, out_channels=n_filters, kernel_size=1)
        self.blocks, self.downs = nn.ModuleList(), nn.ModuleList()
        width = n_filters
        for level in range(n_scales):
            self.blocks.extend(VunetRNB(channels=width, conv_layer=conv_layer, dropout_prob=dropout_prob)
                               for _ in range(RNB_PER_SCALE))
            if level < n_scales - 1:
                wider = min(2 * width, max_filters)
                self.downs.append(Downsample(width, wider))
                width = wider

    def _run(self, image):
        feats, h = [], self.nin(image)
        blocks = iter(self.blocks)
        for level in range(self.n_scales):
            for i in range(RNB_PER_SCALE):
                if i == 0:
                    h = next(blocks)(h)
                else:
                    # the previous block's output has two readers, this block and the skip connection: the skip reads the
                    # alias this block hands back, so both gradients meet in this block's data-gradient epilogue
                    h, feats[-1] = next(blocks)(h, passthrough=True)
                feats.append(h)
            if level < self.n_scales - 1:
                # the skip connection reads the alias the down-sampling layer hands back: one gradient chain, the skip's
                # gradient added inside that layer's data-gradient kernel instead of by an autograd add over the tensor
                h, feats[-1] = self.downs[level](h, passthrough=True)
        return feats


class EncUp(_Pyramid):
    def forward(self, x, **kwargs):
        return self._run(x)


class DecUp(_Pyramid):
    def forward(self, c):
        return self._run(c)


def latent_sample(p, eps: Optional[torch.Tensor] = None):
    """:151-156 -- unit-variance sample around ``p``."""
    if eps is None and p.is_cuda and ops.kernel_noise_enabled():
        return ops.UnitSample.apply(p)   # noise drawn inside the kernel: one launch, identity backward
    return ops.Reparam.apply(p, torch.zeros_like(p), _noise_like(p, eps))


# ------------------------------------------------------------------------------------------------
# bottleneck encoders  (EncDown :159-219 for VunetOrg, EncDownAlter :520-597 for VunetAlter)
# ------------------------------------------------------------------------------------------------
class _Bottleneck(nn.Module):
    """Per latent scale: skip block -> posterior parameters -> sample -> skip block on (feature, sample) -> 2x up."""

    learn_std = False
    second_block_dropout = True   # :548 builds the second block of EncDownAlter without dropout_prob

    def __init__(self, n_filters, nf_in, subpixel_upsampling, n_scales=2, conv_layer=NormConv2d, dropout_prob=0.0):
        super().__init__()
        self.nin = conv_layer(nf_in, n_filters, kernel_size=1)
        self.n_scales, self.n_rnb = n_scales, RNB_PER_SCALE
        self.blocks, self.ups, self.make_latent_params = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        if self.learn_std:
            self.make_logstds = nn.ModuleList()
        w = n_filters
        for _ in range(n_scales):
            self.blocks.append(_skip_block(w, w, p=dropout_prob))
            self.make_latent_params.append(conv_layer(w, w, kernel_size=3, padding=1))
            if self.learn_std:
                self.make_logstds.append(conv_layer(w, w, kernel_size=3, padding=1))
            self.blocks.append(_skip_block(w, 2 * w, p=dropout_prob if self.second_block_dropout else 0.0))
            self.ups.append(Upsample(w, w, subpixel=True))
        self.fin_block = _skip_block(w, w, p=dropout_prob)
        if self.learn_std:
            self.squash = nn.Sigmoid()

    def _encode(self, feats, eps):
        skips = list(feats)                      # consumed from the coarsest end
        hidden, locs, scales, samples = [], [], [], []
        h = self.nin(skips[-1])
        for s in range(self.n_scales):
            h = self.blocks[2 * s](h, skips.pop())
            hidden.append(h)
            # h feeds the posterior's parameter layer(s) AND the next block, the parameters feed the sample AND the KL term:
            # each reader gets a handle of its own, so that the gradients meet in one launch (ops.fan_out) instead of
            # being summed by the engine one aten::add_ at a time -- on 4 x 4 / 8 x 8 maps every launch is latency
            if self.learn_std:
                h_mu, h_sd, h_next = ops.fan_out(h, 3)
            else:
                (h_mu, h_next), h_sd = ops.fan_out(h, 2), None
            loc_kl, loc = ops.fan_out(self.make_latent_params[s](h_mu), 2)
            locs.append(loc_kl)
            noise = None if eps is None else eps[s]
            if self.learn_std:
                log_sd_kl, log_sd = ops.fan_out(self.make_logstds[s].fused(h_sd, out_act=ops.ACT_SIGMOID), 2)   # conv + squash, :574-575
                scales.append(log_sd_kl)
                z = self.reparametrize(loc, log_sd, noise)
            else:
                z = latent_sample(loc, noise)
            z_out, z = ops.fan_out(z, 2)   # (the decoder reads the sample too)
            samples.append(z_out)
            h = self.blocks[2 * s + 1](h_next, (skips.pop(), z))   # cat([g, z]) read as two sources, :210 / :583
            hidden.append(h)
            h = self.ups[s](h)
        hidden.append(self.fin_block(h, skips.pop()))
        return hidden, locs, scales, samples


class EncDown(_Bottleneck):
    def forward(self, gs, eps: Optional[Sequence[torch.Tensor]] = None):
        hidden, qs, _, zs = self._encode(gs, eps)
        return hidden, qs, zs


class EncDownAlter(_Bottleneck):
    learn_std = True
    second_block_dropout = False

    def forward(self, gs, eps: Optional[Sequence[torch.Tensor]] = None):
        hidden, means, log_stds, zs = self._encode(gs, eps)
        return hidden, means, log_stds, zs

    def reparametrize(self, mu, logstd, eps: Optional[torch.Tensor] = None):
        """:594-597 -- eps * exp(logstd) + mu."""
        if eps is None and mu.is_cuda and ops.kernel_noise_enabled():
            return ops.Reparam.apply(mu, logstd, None)   # noise drawn inside the kernel
        return ops.Reparam.apply(mu, logstd, _noise_like(logstd, eps))


# ------------------------------------------------------------------------------------------------
# decoders  (DecDownAlter :264-424, DecDown :600-783)
# ------------------------------------------------------------------------------------------------
def _decoder_width(nf_in, nf_last, n_scales, level):
    return min(nf_in, nf_last * 2 ** (n_scales - (level + 2)))


class DecDownAlter(nn.Module):
    def __init__(self, n_scales, nf_in, nf_last, nf_out, subpixel_upsampling, conv_layer=NormConv2d,
                 n_latent_scales=2, dropout_prob=0.0):
        super().__init__()
        self.n_rnb, self.n_scales, self.n_latent_scales = RNB_PER_SCALE, n_scales, n_latent_scales
        self.nin = conv_layer(nf_in, nf_in, kernel_size=1)
        self.blocks, self.ups, self.auto_blocks = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        self.out_conv = conv_layer(nf_last, nf_out, kernel_size=3, padding=1)
        self.depth_to_space, self.space_to_depth = DepthToSpace(block_size=2), SpaceToDepth(block_size=2)
        w = nf_in
        for level in range(n_scales):
            self.blocks.append(_skip_block(w, w, conv_layer, dropout_prob))
            if level < n_latent_scales:
                self.auto_blocks.append(_skip_block(w, w, conv_layer, dropout_prob))
            self.blocks.append(_skip_block(w, w, conv_layer, dropout_prob))
            if level < n_scales - 1:
                nxt = _decoder_width(nf_in, nf_last, n_scales, level)
                self.ups.append(Upsample(w, nxt, subpixel=bool(subpixel_upsampling) or level < n_latent_scales))
                w = nxt

    def forward(self, gs, zs_posterior, training, prior_eps: Optional[Sequence[torch.Tensor]] = None):
        skips, latents = list(gs), list(zs_posterior)
        h = self.nin(skips[-1])
        for level in range(self.n_scales):
            h = self.blocks[2 * level](h, skips.pop())
            if level < self.n_latent_scales:
                if training:
                    code = latents.pop(0)
                else:   # sample the appearance from the prior, :349-352
                    code = _noise_like(h, None if prior_eps is None else prior_eps[level])
                h = self.auto_blocks[level](h, code)
            h = self.blocks[2 * level + 1](h, skips.pop())
            if level < self.n_scales - 1:
                h = self.ups[level](h)
        assert not skips and not (training and latents)
        return self.out_conv(h)


class DecDown(nn.Module):
    """Decoder of the original VUnet with the 4-group autoregressive prior over SpaceToDepth sub-grids."""

    def __init__(self, n_scales, nf_in, nf_last, nf_out, subpixel_upsampling, conv_layer=NormConv2d,
                 n_latent_scales=2, dropout_prob=0.0):
        super().__init__()
        self.n_rnb, self.n_scales, self.n_latent_scales = RNB_PER_SCALE, n_scales, n_latent_scales
        self.nin = conv_layer(nf_in, nf_in, kernel_size=1)
        self.blocks, self.ups = nn.ModuleList(), nn.ModuleList()
        self.latent_nins, self.auto_lp, self.auto_blocks = nn.ModuleDict(), nn.ModuleDict(), nn.ModuleDict()
        self.out_conv = conv_layer(nf_last, nf_out, kernel_size=3, padding=1)
        self.depth_to_space, self.space_to_depth = DepthToSpace(block_size=2), SpaceToDepth(block_size=2)
        lat, w = nf_in, nf_in
        for level in range(n_scales):
            self.blocks.append(_skip_block(w, w, conv_layer, dropout_prob))
            if level < n_latent_scales:
                key = f"l_{level}"
                self.latent_nins[key] = conv_layer(2 * lat, lat, kernel_size=1)
                self.auto_lp[key] = nn.ModuleList(conv_layer(4 * lat, lat, kernel_size=3, padding=1) for _ in range(4))
                self.auto_blocks[key] = nn.ModuleList(
                    [VunetRNB(channels=lat, dropout_prob=dropout_prob)] +
                    [VunetRNB(channels=4 * lat, a_channels=lat, residual=True, dropout_prob=dropout_prob)
                     for _ in range(3)])
            self.blocks.append(_skip_block(w, w, conv_layer, dropout_prob))
            if level < n_scales - 1:
                nxt = _decoder_width(nf_in, nf_last, n_scales, level)
                self.ups.append(Upsample(w, nxt, subpixel=bool(subpixel_upsampling) or level < n_latent_scales))
                w = nxt

    def _groups(self, x):
        """split the 2x2 sub-grids of x into 4 channel groups (:776-779)"""
        return list(torch.split(self.space_to_depth(x), x.shape[1], dim=1))

    def _ungroup(self, parts):
        return self.depth_to_space(torch.cat(parts, dim=1))

    def forward(self, gs, zs_posterior, training, prior_eps=None):
        skips, latents = list(gs), list(zs_posterior)
        hs, ps, zs = [], [], []
        h = self.nin(skips[-1])
        for level in range(self.n_scales):
            h = self.blocks[2 * level](h, skips.pop())
            hs.append(h)
            if level < self.n_latent_scales:
                key = f"l_{level}"
                teacher = self._groups(latents[0]) if training else None
                feat = self.space_to_depth(self.auto_blocks[key][0](h))
                p_parts, z_parts = [], []
                for g in range(4):
                    p_g = self.auto_lp[key][g](feat)
                    z_g = latent_sample(p_g, None if prior_eps is None else prior_eps[level][g])
                    p_parts.append(p_g)
                    z_parts.append(z_g)
                    fed_back = teacher.pop(0) if training else z_g
                    if g < 3:
                        feat = self.auto_blocks[key][g + 1](feat, fed_back)
                assert not teacher
                ps.append(self._ungroup(p_parts))
                z_prior = self._ungroup(z_parts)
                zs.append(z_prior)
                z = latents.pop(0) if training else z_prior
                h = self.latent_nins[key].fused((h, z))        # cat([h, z]) read as two sources, :756-757
            h = self.blocks[2 * level + 1](h, skips.pop())
            hs.append(h)
            if level < self.n_scales - 1:
                h = self.ups[level](h)
        assert not skips and not (training and latents)
        return self.out_conv(hs[-1]), hs, ps, zs


# ------------------------------------------------------------------------------------------------
# the two generators
# ------------------------------------------------------------------------------------------------
class _VunetBase(nn.Module):
    layer_norm_ok = False
    bottleneck_cls = EncDown
    decoder_cls = DecDown

    def __init__(self, init_fn=None, n_channels_x=3, **kwargs):
        super().__init__()
        self.spatial_size = kwargs["spatial_size"]
        self.n_scales = _count_scales(kwargs)
        # a multi-part appearance input (n_channels_x > 3) arrives at reduced resolution: box_factor fewer scales
        self.n_scales_x = self.n_scales - kwargs["box_factor"] if n_channels_x > 3 else self.n_scales
        self.n_channels_x = n_channels_x
        p_drop = kwargs.get("dropout_prob", 0.0)
        n_lat, nf0, nf1 = kwargs["n_latent_scales"], kwargs["nf_start"], kwargs["nf_max"]
        conv_layer, label = _pick_conv_layer(kwargs, init_fn, self.layer_norm_ok)
        print("Vunet using " + label + " as conv layers.")
        self.eu = EncUp(self.n_scales_x, n_filters=nf0, max_filters=nf1, conv_layer=conv_layer, nf_in=n_channels_x,
                        dropout_prob=p_drop)
        self.ed = self.bottleneck_cls(n_filters=nf1, nf_in=nf1, conv_layer=conv_layer, n_scales=n_lat,
                                      subpixel_upsampling=kwargs["subpixel_upsampling"], dropout_prob=p_drop)
        self.du = DecUp(self.n_scales, n_filters=nf0, max_filters=nf1, conv_layer=conv_layer, dropout_prob=p_drop)
        self.dd = self.decoder_cls(self.n_scales, nf1, nf0, nf_out=3, conv_layer=conv_layer, n_latent_scales=n_lat,
                                   subpixel_upsampling=kwargs["subpixel_upsampling"], dropout_prob=p_drop)

    # ---- pose encoder on a second HIP stream ---------------------------------------------------------------
    # The small-map layers (<= 16x16) of eu and du are latency-bound launches that leave most CUs idle; run side
    # by side they fill each other's gaps.  Autograd replays every node on the stream of its forward, so the
    # backward passes of du and of eu/ed overlap the same way.
    _side_stream = None

    def enable_two_streams(self, on: bool = True):
        self._side_stream = torch.cuda.Stream() if on else None
        return self

    def _fork_pose_encoder(self, c):
        side = self._side_stream
        if side is None or not c.is_cuda:
            return self.du(c)
        main = torch.cuda.current_stream()
        side.wait_stream(main)               # inputs and the step's packed weights were produced on the main stream
        c.record_stream(side)
        with torch.cuda.stream(side):
            return self.du(c)

    def _join_pose_encoder(self, gs):
        side = self._side_stream
        if side is None or not gs[0].is_cuda:
            return gs
        main = torch.cuda.current_stream()
        main.wait_stream(side)
        for g in gs:
            g.record_stream(main)
        return gs

    def join_streams(self):
        """After backward(): gradients the side stream wrote in place must be visible to the optimiser's stream."""
        if self._side_stream is not None:
            torch.cuda.current_stream().wait_stream(self._side_stream)


class VunetAlter(_VunetBase):
    """:426-515 -- the generator ``ShapePoseNet`` trains (learned-variance posterior, no autoregressive prior)."""

    layer_norm_ok = True
    bottleneck_cls = EncDownAlter
    decoder_cls = DecDownAlter

    def forward(self, x, c, eps: Optional[Sequence[torch.Tensor]] = None, after_encoder=None):
        """``after_encoder``: optional callable invoked once the appearance pyramid has been issued and before the
        bottleneck -- the trainer's hook for work that should run beside the latency-bound small-map layers that follow
        (the reference's call is ``forward(x, c)``)."""
        gs = self._fork_pose_encoder(c)      # du(c) shares nothing with eu / ed until dd: issued first, side stream
        hs = self.eu(x)
        if after_encoder is not None:
            after_encoder()
        _, means, logstds, zs = self.ed(hs, eps)
        imgs = self.dd(self._join_pose_encoder(gs), zs, training=True)
        return imgs, means, logstds, (hs, means, logstds)

    def test_forward(self, c, prior_eps: Optional[Sequence[torch.Tensor]] = None):
        return self.dd(self.du(c), [], training=False, prior_eps=prior_eps)

    def transfer(self, x, c, eps: Optional[Sequence[torch.Tensor]] = None):
        return self.transfer_code(self.appearance_code(x, eps), c)   # posterior means as the code, :508-515

    def appearance_code(self, x, eps: Optional[Sequence[torch.Tensor]] = None):
        """The appearance half of ``transfer``: the posterior means of ``ed(eu(x))``."""
        _, means, _, _ = self.ed(self.eu(x), eps)
        return list(means)

    def transfer_code(self, means, c):
        """The pose half of ``transfer``; a batch-1 code is broadcast over the frames of ``c`` (render loop)."""
        n = c.shape[0]
        means = [m if m.shape[0] == n else m.expand(n, -1, -1, -1).contiguous() for m in means]
        return self.dd(self.du(c), means, training=True)


class VunetOrg(_VunetBase):
    """:18-106 -- the original VUnet (``experiments/vunet.py``)."""

    def forward(self, x, c, eps=None, prior_eps=None):
        gs = self._fork_pose_encoder(c)
        hs = self.eu(x)
        _, qs, zs = self.ed(hs, eps)
        gs = self._join_pose_encoder(gs)
        imgs, ds, ps, _ = self.dd(gs, zs, training=True, prior_eps=prior_eps)
        return imgs, qs, ps, (hs, qs, gs, ds)

    def test_forward(self, c, prior_eps=None):
        return self.dd(self.du(c), [], training=False, prior_eps=prior_eps)[0]

    def transfer(self, x, c, eps=None, prior_eps=None):
        _, qs, _ = self.ed(self.eu(x), eps)
        return self.dd(self.du(c), list(qs), training=True, prior_eps=prior_eps)[0]


# ------------------------------------------------------------------------------------------------
# Regressor  (:786-824)
# ------------------------------------------------------------------------------------------------
class Linear(nn.Module):
    """nn.Linear (keys ``weight`` [out, in], ``bias``) on the 1x1 path of the MFMA conv kernel."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        ref = nn.Linear(in_features, out_features)      # default init
        self.weight = nn.Parameter(ref.weight.detach().clone())
        self.bias = nn.Parameter(ref.bias.detach().clone())

    def fused(self, x, out_act=ops.ACT_NONE):
        parts = x if isinstance(x, (tuple, list)) else (x, None)
        x1, x2 = (None if t is None else t.reshape(t.shape[0], -1, 1, 1) for t in parts)
        w = self.weight.reshape(self.out_features, self.in_features, 1, 1)
        return ops.fused_conv(x1, x2, None, w, None, self.bias, None, None, ops.ConvCfg(kind=1, k=1, out_act=out_act))

    def forward(self, x):
        return self.fused(x).reshape(x.shape[0], self.out_features)


class Regressor(nn.Module):
    """Latent -> 2-D key-point probe.  The embedders are full-window valid convolutions (kernel = latent
    width), i.e. linear maps of the flattened latent: they run on the 1x1 path with the weight viewed as
    ``[out, in*k*k, 1, 1]``; state-dict keys and shapes stay the reference's (``embedders.i.weight``: [out, in, k, k])."""

    def __init__(self, n_out, n_latent_scales, nf_max, latent_widths, linear_width_factor, n_linear=2, **kwargs):
        super().__init__()
        self.n_stages, self.n_linear = n_latent_scales, n_linear
        self.linear_width = n_latent_scales * nf_max * linear_width_factor
        self.embedders = nn.ModuleList(Conv2d(nf_max, linear_width_factor * nf_max, kernel_size=latent_widths[i])
                                       for i in range(n_latent_scales))
        self.linears = nn.ModuleList()
        self.act_fn = nn.ReLU()
        lw = self.linear_width
        for i in range(n_linear):
            halve_in = 2 if lw // 2 ** (n_linear - i) > n_out else 1
            halve_out = 2 if lw // 2 ** (n_linear - i - 1) > n_out else 1
            fan_in = lw // halve_in ** i
            self.linears.append(Linear(fan_in, n_out if i == n_linear - 1 else lw // halve_out ** (i + 1)))

    def forward(self, embeddings: list):
        feats = []
        for latent, emb in zip(reversed(embeddings), self.embedders):
            assert latent.shape[-1] == emb.k and latent.shape[-2] == emb.k, "embedder kernel must equal the latent width"
            w = emb.weight.reshape(emb.weight.shape[0], -1, 1, 1)
            feats.append(ops.fused_conv(latent.reshape(latent.shape[0], -1, 1, 1), None, None, w, None, emb.bias, None,
                                        None, ops.ConvCfg(kind=1, k=1, out_act=ops.ACT_RELU)))
        # two embeddings are read as one concatenated source by the first linear layer
        h = tuple(feats) if len(feats) == 2 else (feats[0] if len(feats) == 1 else torch.cat(feats, dim=1))
        last = self.n_linear - 1
        for i, lin in enumerate(self.linears):
            h = lin.fused(h, out_act=ops.ACT_NONE if i == last else ops.ACT_RELU)
        return h.reshape(h.shape[0], -1)
