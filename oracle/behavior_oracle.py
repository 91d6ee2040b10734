"""CPU oracle for the behaviour front half of BASELINE config 5: flow sample -> pose_behavior_rnn decode.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import this module: only ``tests/`` and
``__graft_entry__.smoke()`` use it, and only as the *checker*.

A functional (state-dict driven) restatement in plain PyTorch-CPU fp32 of

* ``UnsupervisedTransformer2`` -> ``UnconditionalFlow2`` -> ``UnconditionalFlatDoubleCouplingFlowBlock2``
  (``ActNorm`` + ``DoubleVectorCouplingBlock2`` + ``Shuffle``), both directions
  (models/flow/simple_flow.py:136-176, models/flow/blocks.py:95-128, :276-319, :531-559, :692-704,
  lib/modules.py:236-257, :260-331);
* ``ResidualBehaviorNet``: ``BEncoder`` (one-layer LSTM over the sequence + the two 1x1 ``NormConv2d`` heads),
  ``ResidualRNNDecoder`` (LSTM cell + ``n_out`` + residual) and ``generate_seq``
  (models/pose_behavior_rnn.py:125-209, :463-534, :538-626).  The decoder's ``rnn_type="gru"`` branch is not
  restated: the reference defines ``n_out`` only inside the LSTM branch (:473-478), so its GRU decoder cannot run;
* the flow stage of BASELINE config 4's training loop: ``FlowLoss`` (lib/losses.py:294-331), ActNorm's data-dependent
  initialisation (lib/modules.py:270-290, :303-305) and the step ``latent_flow(bs.detach())`` -> ``flow_loss`` ->
  ``zero_grad`` / ``backward`` / ``step`` (experiments/behavior_net.py:703-714) with the optimiser of :384-395.  Gradients
  come from torch.autograd over the functions of this file and the update from ``torch.optim.Adam`` -- the third-party
  pieces the reference itself uses for them.  Pinned by ``tests/golden/g10_flow_training.npz``;
* the cVAE stage of the same loop (experiments/behavior_net.py:591-660 with ``get_loss`` :134-149, ``kl_loss``
  lib/losses.py:283-291, the gamma controller :111-116, ``Adam(to_optim, lr_init)`` :324-336), ``use_regressor`` off -- with it
  on the reference's own step raises on torch >= 1.5 (recorded in the fixture).  Pinned by ``tests/golden/g11_cvae_training.npz``.

Every function cites the reference lines it follows (paths relative to the upstream repository root) and takes a
flat ``sd`` mapping with the reference's state-dict key names, so the product modules' ``state_dict()`` can be fed
in unchanged.  Parity pin: ``tests/golden/g9_behavior.npz`` was written by ``tests/golden/make_golden.py`` from the
imported reference classes; ``tests/test_oracle_golden.py`` holds this restatement to it.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# --------------------------------------------------------------------------
# BasicFullyConnectedNet -- lib/modules.py:236-257
# --------------------------------------------------------------------------
def fully_connected_depth(sd: SD, p: str) -> int:
    """Number of hidden Linear layers: ``main`` holds Linear at 0, 2, ..., 2*(depth+1) with LeakyReLU between."""
    n = 0
    while f"{p}.main.{2 * n}.weight" in sd:
        n += 1
    return n - 2


def fully_connected(sd: SD, p: str, x: Tensor, use_tanh: bool) -> Tensor:
    """Linear -> LeakyReLU(0.01) -> depth x (Linear -> LeakyReLU) -> Linear [-> Tanh]   (lib/modules.py:240-257)."""
    depth = fully_connected_depth(sd, p)
    h = x
    for i in range(depth + 2):
        h = F.linear(h, sd[f"{p}.main.{2 * i}.weight"], sd[f"{p}.main.{2 * i}.bias"])
        if i < depth + 1:
            h = F.leaky_relu(h, 0.01)
    return torch.tanh(h) if use_tanh else h


# --------------------------------------------------------------------------
# ActNorm (initialised) -- lib/modules.py:292-331
# --------------------------------------------------------------------------
def actnorm_forward(sd: SD, p: str, x: Tensor) -> Tuple[Tensor, Tensor]:
    """h = scale * (x + loc); logdet = H*W*sum(log|scale|) per sample, H = W = 1 here   (lib/modules.py:307-316)."""
    scale, loc = sd[f"{p}.scale"].reshape(1, -1), sd[f"{p}.loc"].reshape(1, -1)
    h = scale * (x + loc)
    logdet = torch.sum(torch.log(torch.abs(scale))) * torch.ones(x.shape[0])
    return h, logdet


def actnorm_reverse(sd: SD, p: str, y: Tensor) -> Tensor:
    """h = y / scale - loc   (lib/modules.py:320-331)."""
    return y / sd[f"{p}.scale"].reshape(1, -1) - sd[f"{p}.loc"].reshape(1, -1)


def actnorm_initialize(sd: SD, p: str, x: Tensor) -> None:
    """loc = -mean, scale = 1 / (std + 1e-6) per channel over the batch (unbiased std); marks the layer initialised
    (lib/modules.py:270-290, :303-305).  In place on ``sd`` (the values, not the autograd graph)."""
    with torch.no_grad():
        mean, std = x.mean(dim=0), x.std(dim=0)
        sd[f"{p}.loc"].copy_((-mean).reshape(sd[f"{p}.loc"].shape))
        sd[f"{p}.scale"].copy_((1.0 / (std + 1e-6)).reshape(sd[f"{p}.scale"].shape))
        sd[f"{p}.initialized"] = torch.tensor(1, dtype=torch.uint8)


# --------------------------------------------------------------------------
# DoubleVectorCouplingBlock2 -- models/flow/blocks.py:276-319
# --------------------------------------------------------------------------
def _swap_halves(x: Tensor) -> Tensor:
    return torch.cat(torch.chunk(x, 2, dim=1)[::-1], dim=1)


def coupling_forward(sd: SD, p: str, x: Tensor) -> Tuple[Tensor, Tensor]:
    """Two affine half-couplings; the halves change roles before the second   (models/flow/blocks.py:296-309)."""
    logdet = torch.zeros(x.shape[0])
    for i in range(2):
        if i % 2 != 0:
            x = _swap_halves(x)
        xa, xk = torch.chunk(x, 2, dim=1)
        scale = fully_connected(sd, f"{p}.s.{i}", xa, True)
        xk = xk * scale.exp() + fully_connected(sd, f"{p}.t.{i}", xa, False)
        x = torch.cat((xa, xk), dim=1)
        logdet = logdet + scale.reshape(x.shape[0], -1).sum(dim=1)
    return x, logdet


def coupling_reverse(sd: SD, p: str, x: Tensor) -> Tensor:
    """models/flow/blocks.py:310-319."""
    for i in (1, 0):
        if i % 2 == 0:
            x = _swap_halves(x)
        xa, xk = torch.chunk(x, 2, dim=1)
        xk = (xk - fully_connected(sd, f"{p}.t.{i}", xa, False)) * fully_connected(sd, f"{p}.s.{i}", xa, True).neg().exp()
        x = torch.cat((xa, xk), dim=1)
    return x


# --------------------------------------------------------------------------
# UnconditionalFlatDoubleCouplingFlowBlock2 / UnconditionalFlow2 -- models/flow/blocks.py:531-559, :95-128
# --------------------------------------------------------------------------
def flow_n_blocks(sd: SD, p: str = "flow") -> int:
    n = 0
    while f"{p}.sub_layers.{n}.norm_layer.loc" in sd:
        n += 1
    return n


def flow_forward(sd: SD, x: Tensor, p: str = "flow") -> Tuple[Tensor, Tensor]:
    """x [B, C] -> (z [B, C], logdet [B]); ActNorm, coupling, shuffle per block   (models/flow/blocks.py:111-121, :540-551).
    An ActNorm whose ``initialized`` flag is 0 takes its statistics from the batch as it reaches the layer (lib/modules.py:303-305)."""
    logdet = torch.zeros(x.shape[0])
    for i in range(flow_n_blocks(sd, p)):
        q = f"{p}.sub_layers.{i}"
        flag = sd.get(f"{q}.norm_layer.initialized")
        if flag is not None and int(flag) == 0:
            actnorm_initialize(sd, f"{q}.norm_layer", x.detach())
        x, ld = actnorm_forward(sd, f"{q}.norm_layer", x)
        logdet = logdet + ld
        x, ld = coupling_forward(sd, f"{q}.coupling", x)
        logdet = logdet + ld
        x = x[:, sd[f"{q}.shuffle.forward_shuffle_idx"].long()]
    return x, logdet


def flow_reverse(sd: SD, z: Tensor, p: str = "flow") -> Tensor:
    """z [B, C] -> x [B, C]   (models/flow/blocks.py:122-125, :552-557; simple_flow.py:167-170)."""
    x = z
    for i in reversed(range(flow_n_blocks(sd, p))):
        q = f"{p}.sub_layers.{i}"
        x = x[:, sd[f"{q}.shuffle.backward_shuffle_idx"].long()]
        x = coupling_reverse(sd, f"{q}.coupling", x)
        x = actnorm_reverse(sd, f"{q}.norm_layer", x)
    return x


# --------------------------------------------------------------------------
# FlowLoss + the flow stage's optimisation step -- lib/losses.py:294-331, experiments/behavior_net.py:384-395, :703-714
# --------------------------------------------------------------------------
def flow_loss(z: Tensor, logdet: Tensor, noise: Optional[Tensor] = None):
    """``FlowLoss.forward`` on flat z [B, C]: nll = 0.5 sum z^2 per row (lib/losses.py:330-331; the reference sums over
    [1, 2, 3] of [B, C, 1, 1]), loss = mean(nll) - mean(logdet); ``reference_nll_loss`` is the same nll of a standard-normal
    draw (``noise``; logged only).  -> (loss, log)."""
    nll_loss = torch.mean(0.5 * torch.sum(torch.pow(z, 2), dim=1))
    assert logdet.dim() == 1
    nlogdet_loss = -torch.mean(logdet)
    loss = nll_loss + nlogdet_loss
    log = {"flow_loss": float(loss.detach()), "nlogdet_loss": float(nlogdet_loss.detach()), "nll_loss": float(nll_loss.detach())}
    if noise is not None:
        log["reference_nll_loss"] = float(torch.mean(0.5 * torch.sum(torch.pow(noise, 2), dim=1)))
    return loss, log


def flow_parameters(sd: SD, p: str = "flow") -> List[str]:
    """The flow's trainable tensors in ``nn.Module.named_parameters()`` order (per block: ActNorm loc, scale; the s nets'
    then the t nets' Linear weight / bias) -- the order ``torch.optim.Adam``'s state dict indexes them by."""
    names = []
    for i in range(flow_n_blocks(sd, p)):
        q = f"{p}.sub_layers.{i}"
        names += [f"{q}.norm_layer.loc", f"{q}.norm_layer.scale"]
        for kind in ("s", "t"):
            for j in range(2):
                for li in range(fully_connected_depth(sd, f"{q}.coupling.{kind}.{j}") + 2):
                    names += [f"{q}.coupling.{kind}.{j}.main.{2 * li}.weight", f"{q}.coupling.{kind}.{j}.main.{2 * li}.bias"]
    return names


def flow_optimizer(sd: SD, lr: float, weight_decay: float = 0.0, betas=(0.5, 0.9), p: str = "flow"):
    """``Adam(params=[{"params": latent_flow.parameters(), "name": "latent_flow"}], lr=flow_lr * batch_size, betas=(0.5, 0.9),
    weight_decay=...)`` (experiments/behavior_net.py:384-392) over the leaves of ``sd`` (made to require grad)."""
    params = [sd[n].requires_grad_(True) for n in flow_parameters(sd, p)]
    return torch.optim.Adam(params=[{"params": params, "name": "latent_flow"}], lr=lr, betas=betas, weight_decay=weight_decay)


def flow_train_step(sd: SD, opt, bs: Tensor, noise: Optional[Tensor] = None, p: str = "flow") -> Dict[str, float]:
    """experiments/behavior_net.py:703-714: ``gauss, logdet = latent_flow(bs.detach())``; ``flow_loss``; ``zero_grad``;
    ``backward``; ``step``.  -> the step's log."""
    z, logdet = flow_forward(sd, bs.detach(), p)
    loss, log = flow_loss(z, logdet, noise)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return log


# --------------------------------------------------------------------------
# recurrent cell -- torch.nn.LSTMCell / one-layer torch.nn.LSTM as the reference instantiates them
# --------------------------------------------------------------------------
def lstm_cell(w_ih: Tensor, w_hh: Tensor, b_ih: Tensor, b_hh: Tensor, x: Tensor, h: Tensor, c: Tensor):
    """Gate order i, f, g, o (torch.nn.LSTMCell): c' = f*c + i*g, h' = o*tanh(c')."""
    gates = F.linear(x, w_ih, b_ih) + F.linear(h, w_hh, b_hh)
    i, f, g, o = gates.chunk(4, dim=1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    return torch.sigmoid(o) * torch.tanh(c2), c2


# --------------------------------------------------------------------------
# ResidualRNNDecoder / generate_seq -- models/pose_behavior_rnn.py:487-506, :603-626
# --------------------------------------------------------------------------
def generate_seq(sd: SD, b: Tensor, x_pose: Tensor, length: int, start_frame: int, p: str = "decoder"):
    """Roll the residual decoder out from ``x_pose[:, start_frame]`` with hidden = cell = ``b``.

    Returns (xs [B, len, n_kps], cs [B, len, n_kps]): cs holds each step's *input* pose, as the reference returns
    it (``return out + res, res``, :506; "changes are here velocities", :619)."""
    w_ih, w_hh = sd[f"{p}.rnn.weight_ih"], sd[f"{p}.rnn.weight_hh"]
    b_ih, b_hh = sd[f"{p}.rnn.bias_ih"], sd[f"{p}.rnn.bias_hh"]
    x = x_pose[:, start_frame]
    h, c = b, b
    xs: List[Tensor] = []
    cs: List[Tensor] = []
    for _ in range(length):
        res = x
        if f"{p}.n_in.weight" in sd:
            x = F.linear(x, sd[f"{p}.n_in.weight"], sd[f"{p}.n_in.bias"])
        h, c = lstm_cell(w_ih, w_hh, b_ih, b_hh, x, h, c)
        x = F.linear(h, sd[f"{p}.n_out.weight"], sd[f"{p}.n_out.bias"]) + res
        xs.append(x)
        cs.append(res)
    return torch.stack(xs, dim=1), torch.stack(cs, dim=1)


# --------------------------------------------------------------------------
# BEncoder -- models/pose_behavior_rnn.py:175-209
# --------------------------------------------------------------------------
def _norm_linear(sd: SD, p: str, x: Tensor) -> Tensor:
    """``NormConv2d`` with a 1x1 kernel on a 1x1 map: gamma * (g * v/||v|| . x + bias) + beta   (lib/modules.py:135-145)."""
    v, g = sd[f"{p}.conv.weight_v"], sd[f"{p}.conv.weight_g"]
    w = (v * (g / v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, 1, 1, 1))).reshape(v.shape[0], -1)
    y = F.linear(x, w, sd[f"{p}.conv.bias"])
    return sd[f"{p}.gamma"].reshape(1, -1) * y + sd[f"{p}.beta"].reshape(1, -1)


def infer_b(sd: SD, seq: Tensor, eps: Optional[Tensor] = None, sample_noise: Optional[Tensor] = None, p: str = "b_enc"):
    """One-layer LSTM over ``seq`` [B, T, n_kps] from a zero state; ``pre`` = last hidden.  With the information
    bottleneck: mu / logstd heads and b = eps*exp(logstd) + mu, or (``sample=True``) pure noise   (:175-209).

    Returns (b, mu, logstd, pre); without the heads in ``sd``: pre alone."""
    w_ih, w_hh = sd[f"{p}.rnn.weight_ih_l0"], sd[f"{p}.rnn.weight_hh_l0"]
    b_ih, b_hh = sd[f"{p}.rnn.bias_ih_l0"], sd[f"{p}.rnn.bias_hh_l0"]
    bsz, hid = seq.shape[0], w_hh.shape[1]
    h, c = torch.zeros(bsz, hid), torch.zeros(bsz, hid)
    for t in range(seq.shape[1]):
        h, c = lstm_cell(w_ih, w_hh, b_ih, b_hh, seq[:, t], h, c)
    pre = h
    if f"{p}.mu_fn.gamma" not in sd:
        return pre
    mu, logstd = _norm_linear(sd, f"{p}.mu_fn", pre), _norm_linear(sd, f"{p}.std_fn", pre)
    if sample_noise is not None:
        return sample_noise, mu, logstd, pre
    b = (torch.zeros_like(mu) if eps is None else eps) * torch.exp(logstd) + mu
    return b, mu, logstd, pre


def behavior_net_forward(sd: SD, x1: Tensor, x2: Tensor, length: int, start_frame: int = 0,
                         eps: Optional[Tensor] = None, sample_noise: Optional[Tensor] = None):
    """``ResidualBehaviorNet.forward`` with the bottleneck: (xs, cs, b, mu, logstd, pre)   (:574-586)."""
    b, mu, logstd, pre = infer_b(sd, x1, eps, sample_noise)
    xs, cs = generate_seq(sd, b, x2, length, start_frame)
    return xs, cs, b, mu, logstd, pre


# --------------------------------------------------------------------------
# the cVAE stage's optimisation step -- experiments/behavior_net.py:591-660, :111-116, :134-149, :329-336; lib/losses.py:283-291
# --------------------------------------------------------------------------
def kl_loss(mu: Tensor, logstd: Tensor) -> Tensor:
    """lib/losses.py:283-291: mean over the batch of sum(-logstd + 0.5 (std^2 + mu^2)) - 0.5 dim."""
    dim = mu.shape[1]
    std = torch.exp(logstd)
    return (torch.sum(-logstd + 0.5 * (std ** 2 + mu ** 2), dim=-1) - 0.5 * dim).mean()


def behavior_parameters(sd: SD) -> Tuple[List[str], List[str]]:
    """(encoder names, decoder names) in ``named_parameters()`` order: the two Adam param groups "z_enc" and "dec"
    (experiments/behavior_net.py:324-327)."""
    enc = [f"b_enc.rnn.{n}_l0" for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    for h in ("mu_fn", "std_fn"):
        if f"b_enc.{h}.gamma" in sd:
            enc += [f"b_enc.{h}.{n}" for n in ("beta", "gamma", "conv.bias", "conv.weight_g", "conv.weight_v")]
    dec = [f"decoder.rnn.{n}" for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")] + ["decoder.n_out.weight", "decoder.n_out.bias"]
    if "decoder.n_in.weight" in sd:
        dec += ["decoder.n_in.weight", "decoder.n_in.bias"]
    return enc, dec


def behavior_optimizer(sd: SD, lr: float):
    """``Adam([{"params": net.b_enc.parameters(), "name": "z_enc"}, {"params": net.decoder.parameters(), "name": "dec"}],
    lr=lr_init)`` (experiments/behavior_net.py:324-336) over the leaves of ``sd``."""
    enc, dec = behavior_parameters(sd)
    return torch.optim.Adam([{"params": [sd[n].requires_grad_(True) for n in enc], "name": "z_enc"},
                             {"params": [sd[n].requires_grad_(True) for n in dec], "name": "dec"}], lr=lr)


def update_gamma(gamma: float, avg_kl: float, gamma_step: float, imax: float) -> float:
    """``__update_gamma`` (experiments/behavior_net.py:111-116)."""
    return max(gamma - gamma_step * (imax - avg_kl), 0)


def cvae_train_step(sd: SD, opt, kps: Tensor, eps: Tensor, gamma: float, recon_loss_weight: float, gamma_step: float, imax: float):
    """One step of ``train_fn`` with ``only_flow`` and ``use_regressor`` off (experiments/behavior_net.py:591-660):
    ``prepare_input`` (lib/utils.py:914-917), ``net(seq_b, seq_b, seq_len)``, ``get_loss`` (:134-149), ``kl_loss``,
    ``loss = recon_loss_weight * recon + gamma * kl`` (:606-611; ``cvae: False``), ``zero_grad`` / ``backward`` / ``step``, then the
    gamma update.  The second ``net(seq_2, ...)`` pass of the reference (:600-603) feeds nothing and is not restated.
    -> (log, new gamma, (xs, b))."""
    seq_b, target = kps[:, :-1], kps[:, 1:]
    seq_len = seq_b.shape[1]
    xs, cs, b, mu, logstd, pre = behavior_net_forward(sd, seq_b, seq_b, seq_len, 0, eps=eps)
    r = F.mse_loss(xs, target, reduction="none")
    recon, per_seq = torch.mean(r), torch.mean(r, dim=[0, 2])
    kl = kl_loss(mu, logstd)
    loss = recon_loss_weight * recon + gamma * kl
    opt.zero_grad()
    loss.backward()
    opt.step()
    new_gamma = update_gamma(gamma, float(kl.detach()), gamma_step, imax)
    log = {"loss": float(loss.detach()), "loss_recon": float(recon.detach()), "kl_loss": float(kl.detach()), "gamma_used": gamma,
           "gamma": new_gamma, "mu_s": float(mu.detach().mean()), "logstd_s": float(logstd.detach().mean()),
           "loss_per_seq_recon": per_seq.detach()}
    return log, new_gamma, (xs.detach(), b.detach())


# --------------------------------------------------------------------------
# decoded pose vectors -> pixel keypoints -- data/data_conversions_3d.py:178-211, :588-605, :892-912, :1139-1140
# --------------------------------------------------------------------------
def poses_to_keypoints(x, data_mean, data_std, dim_to_ignore, extrinsics, intrinsics, image_size, spatial_size):
    """numpy, as the reference computes it per frame: ``unNormalizeData`` (zeros at the ignored dimensions, then * std +
    mean), ``apply_affine_transform``, ``camera_projection``, joint rescale.  x: [T, n_use] -> [T, J, 2] float64."""
    import numpy as np
    x = np.asarray(x)
    t, d = x.shape[0], data_mean.shape[0]
    orig = np.zeros((t, d), dtype=np.float32)
    use = np.array([i for i in range(d) if i not in dim_to_ignore])
    orig[:, use] = x
    orig = np.multiply(orig, np.repeat(data_std.reshape((1, d)), t, axis=0)) + np.repeat(data_mean.reshape((1, d)), t, axis=0)
    out = []
    size_arr = np.full((1, 2), spatial_size, dtype=float)
    for p in orig.reshape(t, -1, 3):
        x_hom = np.concatenate([p, np.ones((p.shape[0], 1), dtype=p.dtype)], axis=-1)
        pose_c = x_hom @ np.asarray(extrinsics).T
        cam = np.asarray([[intrinsics[0], 0.0, intrinsics[1]], [0.0, intrinsics[2], intrinsics[3]], [0.0, 0.0, 1.0]])
        pose_i = ((pose_c / np.expand_dims(pose_c[..., -1], axis=-1)) @ cam.T)[..., :-1]
        out.append(pose_i * (size_arr / np.expand_dims(np.asarray(image_size, dtype=float), axis=0)))
    return np.stack(out)
