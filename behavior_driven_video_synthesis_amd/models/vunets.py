"""MI355X-native counterparts of the reference's ``models/vunets.py``.

Class names, constructor keywords, public attributes (``eu ed du dd n_scales spatial_size
n_channels_x``), return tuples and state-dict keys follow the reference line by line
(``models/vunets.py:<line>`` citations), so ``experiments.shape_and_pose_net`` can construct and
drive these modules unchanged; all arithmetic runs in the fused gfx950 kernels of ``lib/modules.py``.
Differences that are deliberate and invisible to callers: ``torch.cat`` of skip/latent tensors is
never materialised (the convs read two sources), the reparametrisation is one fused kernel, and the
encoders do not mutate the caller's list.
"""
from __future__ import annotations

from functools import partial
from typing import List, Optional, Sequence

import numpy as np
import torch
from torch import nn
from torch.nn import ModuleList

from .. import ops
from ..lib.modules import (DepthToSpace, Downsample, L2NormConv2d, LayerNormConv2d, NormConv2d, SpaceToDepth,
                           Upsample, VunetRNB, Conv2d)


def _n_scales(kwargs) -> int:
    # models/vunets.py:22-30 / 430-438
    if kwargs["n_scales"] < 6:
        return 1 + int(np.round(np.log2(kwargs["spatial_size"]))) - kwargs["bottleneck_factor"]
    return kwargs["n_scales"]


def _conv_layer(kwargs, init_fn, allow_ln: bool):
    # models/vunets.py:37-44 / 445-453
    t = kwargs["conv_layer_type"]
    if t == "l1":
        return NormConv2d, "L1NormConv2d"
    if t == "l2":
        return partial(L2NormConv2d, init=init_fn, bias=False), "L2NormConv2d"
    if allow_ln:
        return LayerNormConv2d, "LayerNormConv2d"
    raise NotImplementedError("No conv layers others than l1 and l2 normalized ones are available.")


class EncUp(nn.Module):
    """models/vunets.py:109-148 (and DecUp :222-261, identical structure)."""

    def __init__(self, n_scales, n_filters, max_filters, nf_in=3, conv_layer=NormConv2d, dropout_prob=0.0):
        super().__init__()
        self.n_rnb = 2
        self.n_scales = n_scales
        self.nin = conv_layer(in_channels=nf_in, out_channels=n_filters, kernel_size=1)
        self.blocks = nn.ModuleList()
        self.downs = nn.ModuleList()
        nf = n_filters
        for i in range(self.n_scales):
            for _ in range(self.n_rnb):
                self.blocks.append(VunetRNB(channels=nf, conv_layer=conv_layer, dropout_prob=dropout_prob))
            if i + 1 < self.n_scales:
                out_c = min(2 * nf, max_filters)
                self.downs.append(Downsample(nf, out_c))
                nf = out_c

    def forward(self, x, **kwargs):
        hs = []
        h = self.nin(x)
        for i in range(self.n_scales):
            for n in range(self.n_rnb):
                h = self.blocks[2 * i + n](h)
                hs.append(h)
            if i + 1 < self.n_scales:
                h = self.downs[i](h)
        return hs


class DecUp(EncUp):
    """models/vunets.py:222-261."""

    def forward(self, c):
        return super().forward(c)


def latent_sample(p, eps: Optional[torch.Tensor] = None):
    """models/vunets.py:151-156: mean + 1.0 * randn_like(mean)."""
    if eps is None:
        eps = torch.randn_like(p)
    return ops.Reparam.apply(p, torch.zeros_like(p), eps)


class EncDown(nn.Module):
    """models/vunets.py:159-219 (VunetOrg bottleneck, unit-variance posterior)."""

    def __init__(self, n_filters, nf_in, subpixel_upsampling, n_scales=2, conv_layer=NormConv2d, dropout_prob=0.0):
        super().__init__()
        self.nin = conv_layer(nf_in, n_filters, kernel_size=1)
        self.n_scales = n_scales
        self.n_rnb = 2
        self.blocks = nn.ModuleList()
        self.ups = nn.ModuleList()
        self.make_latent_params = nn.ModuleList()
        nf = n_filters
        for i in range(self.n_scales):
            self.blocks.append(VunetRNB(channels=nf, a_channels=nf, residual=True, dropout_prob=dropout_prob))
            self.make_latent_params.append(conv_layer(nf, nf, kernel_size=3, padding=1))
            self.blocks.append(VunetRNB(channels=nf, a_channels=2 * nf, residual=True, dropout_prob=dropout_prob))
            self.ups.append(Upsample(nf, nf, subpixel=True))
        self.fin_block = VunetRNB(channels=nf, a_channels=nf, residual=True, dropout_prob=dropout_prob)

    def forward(self, gs, eps: Optional[Sequence[torch.Tensor]] = None):
        gs = list(gs)
        hs, qs, zs = [], [], []
        h = self.nin(gs[-1])
        for i in range(self.n_scales):
            h = self.blocks[2 * i](h, gs.pop())
            hs.append(h)
            q = self.make_latent_params[i](h)
            qs.append(q)
            z = latent_sample(q, None if eps is None else eps[i])
            zs.append(z)
            h = self.blocks[2 * i + 1](h, (gs.pop(), z))
            hs.append(h)
            h = self.ups[i](h)
        h = self.fin_block(h, gs.pop())
        hs.append(h)
        return hs, qs, zs


class EncDownAlter(nn.Module):
    """models/vunets.py:520-597: posterior ``mu, sigmoid(logstd)`` and reparametrised sample per latent scale."""

    def __init__(self, n_filters, nf_in, subpixel_upsampling, n_scales=2, conv_layer=NormConv2d, dropout_prob=0.0):
        super().__init__()
        self.nin = conv_layer(nf_in, n_filters, kernel_size=1)
        self.n_scales = n_scales
        self.n_rnb = 2
        self.blocks = nn.ModuleList()
        self.ups = nn.ModuleList()
        self.make_latent_params = nn.ModuleList()
        self.make_logstds = nn.ModuleList()
        nf = n_filters
        for i in range(self.n_scales):
            self.blocks.append(VunetRNB(channels=nf, a_channels=nf, residual=True, dropout_prob=dropout_prob))
            self.make_latent_params.append(conv_layer(nf, nf, kernel_size=3, padding=1))
            self.make_logstds.append(conv_layer(nf, nf, kernel_size=3, padding=1))
            # models/vunets.py:548: this block is built without dropout_prob
            self.blocks.append(VunetRNB(channels=nf, a_channels=2 * nf, residual=True))
            self.ups.append(Upsample(nf, nf, subpixel=True))
        self.fin_block = VunetRNB(channels=nf, a_channels=nf, residual=True, dropout_prob=dropout_prob)
        self.squash = nn.Sigmoid()

    def forward(self, gs, eps: Optional[Sequence[torch.Tensor]] = None):
        gs = list(gs)
        hs, means, zs, log_stds = [], [], [], []
        h = self.nin(gs[-1])
        for i in range(self.n_scales):
            h = self.blocks[2 * i](h, gs.pop())
            hs.append(h)
            mu = self.make_latent_params[i](h)
            means.append(mu)
            logstd = self.make_logstds[i].fused(h, out_act=ops.ACT_SIGMOID)  # conv + squash (:574-575)
            log_stds.append(logstd)
            z = self.reparametrize(mu, logstd, None if eps is None else eps[i])
            zs.append(z)
            h = self.blocks[2 * i + 1](h, (gs.pop(), z))  # cat([g, z]) read as two sources (:583)
            hs.append(h)
            h = self.ups[i](h)
        h = self.fin_block(h, gs.pop())
        hs.append(h)
        return hs, means, log_stds, zs

    def reparametrize(self, mu, logstd, eps: Optional[torch.Tensor] = None):
        if eps is None:
            eps = torch.randn_like(logstd)
        return ops.Reparam.apply(mu, logstd, eps)


class DecDownAlter(nn.Module):
    """models/vunets.py:264-424."""

    def __init__(self, n_scales, nf_in, nf_last, nf_out, subpixel_upsampling, conv_layer=NormConv2d,
                 n_latent_scales=2, dropout_prob=0.0):
        super().__init__()
        self.n_rnb = 2
        self.n_scales = n_scales
        self.n_latent_scales = n_latent_scales
        self.nin = conv_layer(nf_in, nf_in, kernel_size=1)
        self.blocks = nn.ModuleList()
        self.ups = nn.ModuleList()
        self.auto_blocks = nn.ModuleList()
        self.out_conv = conv_layer(nf_last, nf_out, kernel_size=3, padding=1)
        self.depth_to_space = DepthToSpace(block_size=2)
        self.space_to_depth = SpaceToDepth(block_size=2)
        nf = nf_in
        for i in range(self.n_scales):
            self.blocks.append(VunetRNB(channels=nf, a_channels=nf, residual=True, conv_layer=conv_layer,
                                        dropout_prob=dropout_prob))
            if i < self.n_latent_scales:
                self.auto_blocks.append(VunetRNB(channels=nf, a_channels=nf, residual=True, conv_layer=conv_layer,
                                                 dropout_prob=dropout_prob))
            self.blocks.append(VunetRNB(channels=nf, a_channels=nf, residual=True, conv_layer=conv_layer,
                                        dropout_prob=dropout_prob))
            if i + 1 < self.n_scales:
                out_c = min(nf_in, nf_last * 2 ** (n_scales - (i + 2)))
                subpixel = True if subpixel_upsampling else (i < self.n_latent_scales)
                self.ups.append(Upsample(nf, out_c, subpixel=subpixel))
                nf = out_c

    def forward(self, gs, zs_posterior, training, prior_eps: Optional[Sequence[torch.Tensor]] = None):
        gs = list(gs)
        zs_posterior = list(zs_posterior)
        h = self.nin(gs[-1])
        lat_count = 0
        for i in range(self.n_scales):
            h = self.blocks[2 * i](h, gs.pop())
            if i < self.n_latent_scales:
                if training:
                    from_dist = zs_posterior.pop(0)
                elif prior_eps is not None:
                    from_dist = prior_eps[lat_count]
                else:
                    from_dist = torch.randn_like(h)
                h = self.auto_blocks[lat_count](h, from_dist)
                lat_count += 1
            h = self.blocks[2 * i + 1](h, gs.pop())
            if i + 1 < self.n_scales:
                h = self.ups[i](h)
        assert not gs
        if training:
            assert not zs_posterior
        return self.out_conv(h)


class VunetAlter(nn.Module):
    """models/vunets.py:426-515."""

    def __init__(self, init_fn=None, n_channels_x=3, **kwargs):
        super().__init__()
        self.spatial_size = kwargs["spatial_size"]
        self.n_scales = _n_scales(kwargs)
        self.n_scales_x = self.n_scales - kwargs["box_factor"] if n_channels_x > 3 else self.n_scales
        self.n_channels_x = n_channels_x
        n_latent_scales = kwargs["n_latent_scales"]
        dropout_prob = kwargs["dropout_prob"] if "dropout_prob" in kwargs else 0.0
        conv_layer, conv_t = _conv_layer(kwargs, init_fn, allow_ln=True)
        print("Vunet using " + conv_t + " as conv layers.")
        self.eu = EncUp(self.n_scales_x, n_filters=kwargs["nf_start"], max_filters=kwargs["nf_max"],
                        conv_layer=conv_layer, nf_in=n_channels_x, dropout_prob=dropout_prob)
        self.ed = EncDownAlter(n_filters=kwargs["nf_max"], nf_in=kwargs["nf_max"], conv_layer=conv_layer,
                               n_scales=n_latent_scales, subpixel_upsampling=kwargs["subpixel_upsampling"],
                               dropout_prob=dropout_prob)
        self.du = DecUp(self.n_scales, n_filters=kwargs["nf_start"], max_filters=kwargs["nf_max"],
                        conv_layer=conv_layer, dropout_prob=dropout_prob)
        self.dd = DecDownAlter(self.n_scales, kwargs["nf_max"], kwargs["nf_start"], nf_out=3, conv_layer=conv_layer,
                               n_latent_scales=n_latent_scales, subpixel_upsampling=kwargs["subpixel_upsampling"],
                               dropout_prob=dropout_prob)

    def forward(self, x, c, eps: Optional[Sequence[torch.Tensor]] = None):
        hs = self.eu(x)
        es, means, logstds, zs_posterior = self.ed(hs, eps)
        gs = self.du(c)
        imgs = self.dd(gs, zs_posterior, training=True)
        activations = hs, means, logstds
        return imgs, means, logstds, activations

    def test_forward(self, c, prior_eps: Optional[Sequence[torch.Tensor]] = None):
        gs = self.du(c)
        return self.dd(gs, [], training=False, prior_eps=prior_eps)

    def transfer(self, x, c, eps: Optional[Sequence[torch.Tensor]] = None):
        hs = self.eu(x)
        es, means, logstds, zs_posterior = self.ed(hs, eps)
        gs = self.du(c)
        return self.dd(gs, list(means), training=True)


class DecDown(nn.Module):
    """models/vunets.py:600-783: decoder with the 4-group autoregressive prior of the original VUnet."""

    def __init__(self, n_scales, nf_in, nf_last, nf_out, subpixel_upsampling, conv_layer=NormConv2d,
                 n_latent_scales=2, dropout_prob=0.0):
        super().__init__()
        self.n_rnb = 2
        self.n_scales = n_scales
        self.n_latent_scales = n_latent_scales
        self.nin = conv_layer(nf_in, nf_in, kernel_size=1)
        self.blocks = nn.ModuleList()
        self.ups = nn.ModuleList()
        self.latent_nins = nn.ModuleDict()
        self.auto_lp = nn.ModuleDict()
        self.auto_blocks = nn.ModuleDict()
        self.out_conv = conv_layer(nf_last, nf_out, kernel_size=3, padding=1)
        self.depth_to_space = DepthToSpace(block_size=2)
        self.space_to_depth = SpaceToDepth(block_size=2)
        nfl = nf_in
        nf = nf_in
        for i in range(self.n_scales):
            self.blocks.append(VunetRNB(channels=nf, a_channels=nf, residual=True, conv_layer=conv_layer,
                                        dropout_prob=dropout_prob))
            if i < self.n_latent_scales:
                scale = f"l_{i}"
                self.latent_nins.update({scale: conv_layer(nfl * 2, nfl, kernel_size=1)})
                clp, cb = ModuleList(), ModuleList()
                for l in range(4):
                    clp.append(conv_layer(4 * nfl, nfl, kernel_size=3, padding=1))
                    if l == 0:
                        cb.append(VunetRNB(channels=nfl, dropout_prob=dropout_prob))
                    else:
                        cb.append(VunetRNB(channels=4 * nfl, a_channels=nfl, residual=True,
                                           dropout_prob=dropout_prob))
                self.auto_lp.update({scale: clp})
                self.auto_blocks.update({scale: cb})
            self.blocks.append(VunetRNB(channels=nf, a_channels=nf, residual=True, conv_layer=conv_layer,
                                        dropout_prob=dropout_prob))
            if i + 1 < self.n_scales:
                out_c = min(nf_in, nf_last * 2 ** (n_scales - (i + 2)))
                subpixel = True if subpixel_upsampling else (i < self.n_latent_scales)
                self.ups.append(Upsample(nf, out_c, subpixel=subpixel))
                nf = out_c

    def _split_groups(self, x):
        return list(torch.split(self.space_to_depth(x), x.shape[1], dim=1))

    def _merge_groups(self, xs):
        return self.depth_to_space(torch.cat(xs, dim=1))

    def forward(self, gs, zs_posterior, training, prior_eps=None):
        gs = list(gs)
        zs_posterior = list(zs_posterior)
        hs, ps, zs = [], [], []
        h = self.nin(gs[-1])
        for i in range(self.n_scales):
            h = self.blocks[2 * i](h, gs.pop())
            hs.append(h)
            if i < self.n_latent_scales:
                scale = f"l_{i}"
                if training:
                    zs_posterior_groups = self._split_groups(zs_posterior[0])
                p_groups, z_groups = [], []
                pre = self.auto_blocks[scale][0](h)
                p_features = self.space_to_depth(pre)
                for l in range(4):
                    p_group = self.auto_lp[scale][l](p_features)
                    p_groups.append(p_group)
                    z_group = latent_sample(p_group, None if prior_eps is None else prior_eps[i][l])
                    z_groups.append(z_group)
                    feedback = zs_posterior_groups.pop(0) if training else z_group
                    if l + 1 < 4:
                        p_features = self.auto_blocks[scale][l + 1](p_features, feedback)
                if training:
                    assert not zs_posterior_groups
                ps.append(self._merge_groups(p_groups))
                z_prior = self._merge_groups(z_groups)
                zs.append(z_prior)
                z = zs_posterior.pop(0) if training else z_prior
                h = self.latent_nins[scale].fused((h, z))  # cat([h, z]) read as two sources (:756-757)
                h = self.blocks[2 * i + 1](h, gs.pop())
                hs.append(h)
            else:
                h = self.blocks[2 * i + 1](h, gs.pop())
                hs.append(h)
            if i + 1 < self.n_scales:
                h = self.ups[i](h)
        assert not gs
        if training:
            assert not zs_posterior
        params = self.out_conv(hs[-1])
        return params, hs, ps, zs


class VunetOrg(nn.Module):
    """models/vunets.py:18-106."""

    def __init__(self, init_fn=None, n_channels_x=3, **kwargs):
        super().__init__()
        self.spatial_size = kwargs["spatial_size"]
        self.n_scales = _n_scales(kwargs)
        self.n_scales_x = self.n_scales - kwargs["box_factor"] if n_channels_x > 3 else self.n_scales
        dropout_prob = kwargs["dropout_prob"] if "dropout_prob" in kwargs else 0.0
        self.n_channels_x = n_channels_x
        n_latent_scales = kwargs["n_latent_scales"]
        conv_layer, conv_t = _conv_layer(kwargs, init_fn, allow_ln=False)
        print("Vunet using " + conv_t + " as conv layers.")
        self.eu = EncUp(self.n_scales_x, n_filters=kwargs["nf_start"], max_filters=kwargs["nf_max"],
                        conv_layer=conv_layer, nf_in=self.n_channels_x, dropout_prob=dropout_prob)
        self.ed = EncDown(n_filters=kwargs["nf_max"], nf_in=kwargs["nf_max"], conv_layer=conv_layer,
                          n_scales=n_latent_scales, subpixel_upsampling=kwargs["subpixel_upsampling"],
                          dropout_prob=dropout_prob)
        self.du = DecUp(self.n_scales, n_filters=kwargs["nf_start"], max_filters=kwargs["nf_max"],
                        conv_layer=conv_layer, dropout_prob=dropout_prob)
        self.dd = DecDown(self.n_scales, kwargs["nf_max"], kwargs["nf_start"], nf_out=3, conv_layer=conv_layer,
                          n_latent_scales=n_latent_scales, subpixel_upsampling=kwargs["subpixel_upsampling"],
                          dropout_prob=dropout_prob)

    def forward(self, x, c, eps=None, prior_eps=None):
        hs = self.eu(x)
        es, qs, zs_posterior = self.ed(hs, eps)
        gs = self.du(c)
        imgs, ds, ps, zs_prior = self.dd(gs, zs_posterior, training=True, prior_eps=prior_eps)
        activations = hs, qs, gs, ds
        return imgs, qs, ps, activations

    def test_forward(self, c, prior_eps=None):
        gs = self.du(c)
        imgs, ds, ps, zs_prior = self.dd(gs, [], training=False, prior_eps=prior_eps)
        return imgs

    def transfer(self, x, c, eps=None, prior_eps=None):
        hs = self.eu(x)
        es, qs, zs_posterior = self.ed(hs, eps)
        gs = self.du(c)
        imgs, _, _, _ = self.dd(gs, list(qs), training=True, prior_eps=prior_eps)
        return imgs


class Regressor(nn.Module):
    """models/vunets.py:786-824: latent -> 2-D keypoints probe.

    The embedders are full-window valid convolutions (kernel = latent width), i.e. linear maps of the
    flattened latent; they run on the 1x1 path of the MFMA conv kernel with the weight viewed as
    ``[out, in*k*k, 1, 1]``.  State-dict keys/shape are the reference's (``embedders.i.weight`` is
    ``[out, in, k, k]``).
    """

    def __init__(self, n_out, n_latent_scales, nf_max, latent_widths, linear_width_factor, n_linear=2, **kwargs):
        super().__init__()
        self.n_stages = n_latent_scales
        self.n_linear = n_linear
        self.linear_width = self.n_stages * nf_max * linear_width_factor
        self.embedders = nn.ModuleList()
        self.linears = nn.ModuleList()
        self.act_fn = nn.ReLU()
        for i in range(self.n_stages):
            self.embedders.append(Conv2d(nf_max, linear_width_factor * nf_max, kernel_size=latent_widths[i]))
        for i in range(self.n_linear):
            arg_in = 2 if self.linear_width // 2 ** (self.n_linear - i) > n_out else 1
            arg_out = 2 if self.linear_width // 2 ** (self.n_linear - i - 1) > n_out else 1
            if i == n_linear - 1:
                self.linears.append(Linear(self.linear_width // arg_in ** i, n_out))
            else:
                self.linears.append(Linear(self.linear_width // arg_in ** i, self.linear_width // arg_out ** (i + 1)))

    def forward(self, embeddings: list):
        outs = []
        for e, emb in zip(reversed(embeddings), self.embedders):
            n = e.shape[0]
            assert e.shape[2] == emb.k and e.shape[3] == emb.k, "embedder kernel must equal the latent width"
            cfg = ops.ConvCfg(kind=1, k=1, out_act=ops.ACT_RELU)
            y = ops.fused_conv(e.reshape(n, -1, 1, 1), None, None, emb.weight.reshape(emb.weight.shape[0], -1, 1, 1),
                               None, emb.bias, None, None, cfg)
            outs.append(y)
        # the two embeddings are read as one concatenated source by the first linear layer
        h = tuple(outs) if len(outs) == 2 else (outs[0] if len(outs) == 1 else torch.cat(outs, dim=1))
        for i in range(self.n_linear):
            h = self.linears[i].fused(h, out_act=ops.ACT_RELU if i < self.n_linear - 1 else ops.ACT_NONE)
        return h.reshape(h.shape[0], -1)


class Linear(nn.Module):
    """nn.Linear (keys weight [out,in], bias) on the 1x1 path of the MFMA conv kernel."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        lin = nn.Linear(in_features, out_features)
        self.weight = nn.Parameter(lin.weight.detach().clone())
        self.bias = nn.Parameter(lin.bias.detach().clone())

    def fused(self, x, out_act=ops.ACT_NONE):
        if isinstance(x, (tuple, list)):
            x1, x2 = (t.reshape(t.shape[0], -1, 1, 1) for t in x)
        else:
            x1, x2 = x.reshape(x.shape[0], -1, 1, 1), None
        cfg = ops.ConvCfg(kind=1, k=1, out_act=out_act)
        return ops.fused_conv(x1, x2, None, self.weight.reshape(self.out_features, self.in_features, 1, 1), None,
                              self.bias, None, None, cfg)

    def forward(self, x):
        return self.fused(x).reshape(x.shape[0], self.out_features)
