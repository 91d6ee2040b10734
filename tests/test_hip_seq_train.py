"""BASELINE config 4 on the GPU, flow stage (csrc/seq_train.hip through the C ABI): the maximum-likelihood step of
experiments/behavior_net.py:703-714 -- forward, FlowLoss, backward, Adam -- against the trajectories the reference's own
modules + FlowLoss + torch.optim.Adam wrote (tests/golden/g10_flow_training.npz) and against the pinned oracle
(oracle/behavior_oracle.py) at the reference configuration's width."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from synth import seeded_randn, synth_behavior_state

pytestmark = pytest.mark.gpu


def _rel(a, b):
    """max |a - b| relative to max |b| (gradient-sized tensors: small entries carry the summation noise of the large ones)."""
    a = a.detach().double().cpu() if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a)).double()
    b = b.detach().double().cpu() if isinstance(b, torch.Tensor) else torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _rel_q(a, b):
    """(the median, the 99th percentile, the maximum) of |a - b| relative to max |b|."""
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    d = (a - b).abs() / b.abs().max().clamp_min(1e-30)
    if d.numel() < 2:
        return float(d.max()), float(d.max()), float(d.max())
    return (float(torch.kthvalue(d, max(1, d.numel() // 2)).values), float(torch.kthvalue(d, max(1, int(0.99 * d.numel()))).values),
            float(d.max()))


def _g10_state(info, arr, tag, seed):
    stored = {k[len(tag) + 4:]: torch.from_numpy(v) for k, v in arr.items() if k.startswith(tag + ".sd.")}
    sd = synth_behavior_state(info["shapes"], seed, stored)
    if info["fresh"]:
        for k in list(sd):
            leaf = k.rsplit(".", 1)[-1]
            if leaf == "initialized":
                sd[k] = torch.tensor(0, dtype=torch.uint8)
            elif leaf == "loc":
                sd[k] = torch.zeros_like(sd[k])
            elif leaf == "scale" and ".norm_layer." in k:
                sd[k] = torch.ones_like(sd[k])
    return sd


def _flow(kw, sd):
    from behavior_driven_video_synthesis_amd.models.flow.simple_flow import UnsupervisedTransformer2
    flow = UnsupervisedTransformer2(**kw)
    flow.load_state_dict(sd)
    return flow.cuda()


def _check_final(flow, tag, info, arr, tol):
    fin = {k: v.detach().cpu() for k, v in flow.state_dict().items()}
    worst = 0.0
    for k, v in arr.items():
        if k.startswith(f"{tag}.final."):
            worst = max(worst, _rel(fin[k[len(tag) + 7:]], v))
    assert worst <= tol, f"parameters after the last step: {worst:.2e} of max|.|"
    for k, (s, a) in info["checksums"].items():
        assert abs(float(fin[k].double().abs().sum()) - a) <= 1e-4 * a + 1e-6, k
    return worst


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("tag", ["even", "odd"])
def test_fused_flow_step_follows_the_reference_trajectory(tag, graph):
    """``FlowTrainEngine.train_step`` (dW fused into Adam, one recorded graph per batch size) for the fixture's three steps:
    every step's FlowLoss log, then parameters and Adam moments.  ``even`` is a fresh flow: ActNorm initialises itself from
    the first batch inside step 1, as in the reference's run."""
    meta, arr = load_golden("g10_flow_training")
    seed, info = meta["seed"], meta["cases"][tag]
    flow = _flow(info["kw"], _g10_state(info, arr, tag, seed))
    eng = flow.flow.train_engine(lr=info["lr"], betas=(0.5, 0.9), weight_decay=info["weight_decay"])
    eng.graph.enabled = graph
    chan, bsz = info["kw"]["flow_in_channels"], info["batch"]
    for it in range(meta["steps"]):
        bs = (0.8 * seeded_randn(f"flowtrain.{tag}.b{it}", (bsz, chan), seed) + 0.3).cuda()
        noise = seeded_randn(f"flowtrain.{tag}.s{it}.eps0", (bsz, chan, 1, 1), seed).reshape(bsz, chan).cuda()
        got = eng.train_step(bs, noise).tolist()
        want = info["logs"][it]
        for name, g in zip(("flow_loss", "reference_nll_loss", "nlogdet_loss", "nll_loss"), got):
            assert abs(g - want[name]) <= 2e-4 * abs(want[name]) + 2e-4, (it, name, g, want[name])
    worst = _check_final(flow, tag, info, arr, 1e-5)      # measured 5.9e-7 (even), 7.1e-8 (odd)
    st = eng.optimizer_state_dict()
    names = [n for n, _ in flow.named_parameters()]
    assert int(st["state"][0]["step"]) == info["adam_step"] and st["param_groups"][0]["name"] == "latent_flow"
    for k, v in arr.items():
        for kind in ("exp_avg", "exp_avg_sq"):
            if k.startswith(f"{tag}.{kind}."):
                assert _rel(st["state"][names.index(k[len(tag) + len(kind) + 2:])][kind], v) <= 2e-5, k
    print(f"\n[g10 {tag} graph={graph}] parameters after 3 fused steps: {worst:.2e} of max|.|")


@pytest.mark.parametrize("tag", ["even", "odd"])
def test_flow_autograd_with_torch_adam_follows_the_reference_trajectory(tag):
    """What an unchanged ``train_fn`` does through the drop-in: ``latent_flow(bs.detach())`` under autograd, the loss in
    torch, ``backward()`` into the HIP backward pass (gradients written out), ``torch.optim.Adam.step()``."""
    meta, arr = load_golden("g10_flow_training")
    seed, info = meta["seed"], meta["cases"][tag]
    flow = _flow(info["kw"], _g10_state(info, arr, tag, seed)).train()
    opt = torch.optim.Adam(params=[{"params": flow.parameters(), "name": "latent_flow"}], lr=info["lr"], betas=(0.5, 0.9),
                           weight_decay=info["weight_decay"])
    chan, bsz = info["kw"]["flow_in_channels"], info["batch"]
    for it in range(meta["steps"]):
        bs = (0.8 * seeded_randn(f"flowtrain.{tag}.b{it}", (bsz, chan), seed) + 0.3).cuda()
        gauss, logdet = flow(bs.detach())
        assert gauss.shape == (bsz, chan, 1, 1) and gauss.requires_grad and logdet.requires_grad
        nll = torch.mean(0.5 * torch.sum(torch.pow(gauss, 2), dim=[1, 2, 3]))     # lib/losses.py:300-305, :330-331
        loss = nll - torch.mean(logdet)
        opt.zero_grad()
        loss.backward()
        opt.step()
        want = info["logs"][it]
        assert abs(float(loss.detach()) - want["flow_loss"]) <= 2e-4 * abs(want["flow_loss"]) + 2e-4, (it, float(loss), want["flow_loss"])
    _check_final(flow, tag, info, arr, 1e-5)


def _random_flow(chan, mid, depth, n_flows, seed, s_gain=0.1):
    from behavior_driven_video_synthesis_amd.models.flow.simple_flow import UnsupervisedTransformer2
    torch.manual_seed(seed)
    flow = UnsupervisedTransformer2(flow_in_channels=chan, flow_mid_channels=mid, flow_hidden_depth=depth, n_flows=n_flows)
    sd = flow.state_dict()
    stored = {k: v for k, v in sd.items() if k.endswith("_shuffle_idx")}
    sd = synth_behavior_state({k: list(v.shape) for k, v in sd.items()}, seed, stored)
    last = f".main.{2 * (depth + 1)}."
    for k in sd:
        if ".coupling.s." in k and last in k:
            sd[k] = sd[k] * s_gain
    flow.load_state_dict(sd)
    return flow.cuda(), sd


@pytest.mark.parametrize("bsz", [1, 16, 17, 48, 64])
def test_flow_gradients_vs_oracle_over_batch_sizes(bsz):
    """Every batch-tile count of the backward kernels: d loss / d x and all parameter gradients of one pass vs torch.autograd
    over the oracle, with a loss that has non-trivial d / d z and d / d logdet."""
    from oracle import behavior_oracle as B
    flow, sd = _random_flow(96, 160, 2, 2, 5)
    x = seeded_randn("tr.x", (bsz, 96), 5)
    wz = seeded_randn("tr.wz", (bsz, 96), 5)
    wl = seeded_randn("tr.wl", (bsz,), 5)
    xg = x.cuda().requires_grad_(True)
    z, logdet = flow(xg)
    (z.reshape(bsz, 96) * wz.cuda()).sum().add((logdet * wl.cuda()).sum()).backward()
    ref = {k: v.clone().requires_grad_(v.dtype.is_floating_point and v.dim() > 0) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    zr, lr_ = B.flow_forward(ref, xr)
    ((zr * wz).sum() + (lr_ * wl).sum()).backward()
    assert _rel(z.reshape(bsz, 96), zr) <= 2e-5 and _rel(logdet, lr_) <= 2e-5
    assert _rel(xg.grad, xr.grad) <= 5e-5
    worst = 0.0
    for n, p in flow.named_parameters():
        assert p.grad is not None, n
        worst = max(worst, _rel(p.grad, ref[n].grad))
    assert worst <= 1e-4, f"{worst:.2e}"


def _away_from_the_kink(sd, depth):
    """Hidden biases of +-6 (alternating units): every LeakyReLU pre-activation of the synthetic flow (standard deviation about
    1.2) then sits at least 3 away from zero, half of the units on each slope.  Without this, of a step's 4.7 M hidden
    pre-activations a few lie within summation rounding of zero, and there the derivative is 1 on one side of a comparison and
    0.01 on the other: one sample's share of one row of dW, and a small change of that sample's signal in everything upstream
    (measured: 6e-2 of max on one row of 2048, 1e-4 on the ActNorm gradients, tools/dbg_flow_train.py) -- which says nothing
    about either implementation.  The generic case is the small-size gradient tests' subject."""
    for k in list(sd):
        if k.endswith(".bias") and ".coupling." in k and f".main.{2 * (depth + 1)}." not in k:
            n = sd[k].numel()
            sd[k] = 6.0 * (1.0 - 2.0 * (torch.arange(n) % 2).float()) + 0.1 * sd[k]
        elif k.endswith(".weight") and ".coupling." in k and f".main.{2 * (depth + 1)}." in k:
            sd[k] = 0.02 * sd[k]          # (hidden activations of size 6 instead of 1: keep the heads' outputs O(1))
    return sd


def test_flow_step_at_the_reference_width_vs_oracle():
    """config/behavior_net.yaml's sizes (1024 channels, 2048 hidden, depth 2, batch 64), 3 of the 15 blocks, three fused steps.
    Graph replay equals eager issue bit for bit; the parameters are updated in place (no padded copies at these sizes).  Every
    step is held to the oracle's step FROM THE SAME STATE (weights, moments and step count copied out of the engine through
    ``optimizer_state_dict`` into ``torch.optim.Adam``): the step's losses, Adam's moments (linear / quadratic in the gradients)
    at the maximum over every tensor's elements, and the weights."""
    from oracle import behavior_oracle as B
    lr = 4.5e-7 * 64            # flow_lr * batch_size (experiments/behavior_net.py:382)
    runs = {}
    for graph in (False, True):
        flow, sd = _random_flow(1024, 2048, 2, 3, 7)
        sd = _away_from_the_kink(sd, 2)
        flow.load_state_dict(sd)
        eng = flow.flow.train_engine(lr=lr, betas=(0.5, 0.9), weight_decay=0.0)
        eng.graph.enabled = graph
        logs, worst = [], dict(loss=0.0, exp_avg=0.0, exp_avg_sq=0.0, w_far=0.0, w_max=0.0)
        for it in range(3):
            batch = seeded_randn(f"w.b{it}", (64, 1024), 7)
            if graph:   # the oracle takes this step from the engine's state
                ref = {k: v.detach().cpu().clone() for k, v in flow.state_dict().items()}
                opt = B.flow_optimizer(ref, lr, 0.0)
                if it:
                    opt.load_state_dict(eng.optimizer_state_dict())
                log = B.flow_train_step(ref, opt, batch)
            logs.append(eng.train_step(batch.cuda(), torch.zeros(64, 1024, device="cuda")).tolist())
            if not graph:
                continue
            for gi, name in ((0, "flow_loss"), (3, "nll_loss"), (2, "nlogdet_loss")):
                worst["loss"] = max(worst["loss"], abs(logs[-1][gi] - log[name]) / abs(log[name]))
            mine, theirs = eng.optimizer_state_dict()["state"], opt.state_dict()["state"]
            assert int(mine[0]["step"]) == int(theirs[0]["step"]) == it + 1
            for i in range(len(theirs)):
                worst["exp_avg"] = max(worst["exp_avg"], _rel(mine[i]["exp_avg"], theirs[i]["exp_avg"]))
                worst["exp_avg_sq"] = max(worst["exp_avg_sq"], _rel(mine[i]["exp_avg_sq"], theirs[i]["exp_avg_sq"]))
            # weights move by lr m / (sqrt(v) + eps) ~ +-lr whatever the gradient's size: an element whose gradient is
            # summation noise may go the other way (at most 2 lr apart) -- bounded, and counted
            far = total = 0
            for k, v in flow.state_dict().items():
                if v.dtype.is_floating_point and v.dim() > 0:
                    d = (v.detach().cpu() - ref[k].detach()).abs()
                    worst["w_max"] = max(worst["w_max"], float(d.max()) / lr)
                    far += int((d > 0.02 * lr).sum())
                    total += d.numel()
            worst["w_far"] = max(worst["w_far"], far / total)
        assert all(lay.w_inplace and lay.b_inplace for lay in eng._all_layers())
        runs[graph] = (logs, {k: v.detach().clone() for k, v in flow.state_dict().items()})
    for k, v in runs[False][1].items():
        assert torch.equal(v, runs[True][1][k]), k
    assert runs[False][0] == runs[True][0]
    print(f"\n[1024/2048 x 3 blocks, 3 steps, each vs the oracle's step from the same state] losses {worst['loss']:.1e}; exp_avg "
          f"{worst['exp_avg']:.1e}, exp_avg_sq {worst['exp_avg_sq']:.1e} of max|.| (maximum over all elements); weights: max |diff| "
          f"{worst['w_max']:.2f} lr, share more than 0.02 lr apart {worst['w_far']:.1e}")
    # measured: 2.3e-7, 6.2e-7, 1.0e-6; 1.94 lr, 9.1e-7
    assert worst["loss"] <= 2e-6 and worst["exp_avg"] <= 1e-5 and worst["exp_avg_sq"] <= 1e-5
    assert worst["w_max"] <= 2.002 and worst["w_far"] <= 1e-5


@pytest.mark.parametrize("shape", [(96, 160, 2, 2, 17), (1024, 2048, 2, 2, 64)])
def test_tile_major_activations_in_the_training_forward_change_nothing(shape):
    """The training forward hands the hidden activations on tile-major as well (include/vunet_seq_tiled.h, layouts 4 / 6 / 2 with a
    row-major copy for the backward pass): two fused steps are bit-identical with and without."""
    chan, mid, depth, n_flows, bsz = shape
    res = {}
    for tiled in (True, False):
        flow, _ = _random_flow(chan, mid, depth, n_flows, 41)
        eng = flow.flow.train_engine(lr=1e-4, betas=(0.5, 0.9))
        eng.tile_activations = tiled
        eng.graph.enabled = False
        logs = [eng.train_step(seeded_randn(f"ta.b{it}", (bsz, chan), 41).cuda(), torch.zeros(bsz, chan, device="cuda")).tolist()
                for it in range(2)]
        res[tiled] = (logs, {k: v.detach().clone() for k, v in flow.state_dict().items()})
    assert res[True][0] == res[False][0]
    for k, v in res[True][1].items():
        assert torch.equal(v, res[False][1][k]), k


@pytest.mark.parametrize("shape", [(96, 160, 2, 2, 17), (96, 160, 3, 2, 48), (1024, 2048, 2, 2, 64), (1024, 2048, 2, 2, 16)])
def test_slab_sum_inside_the_input_gradient_launch_changes_nothing(shape):
    """``vunet_seq_dx_finish``: the last workgroup of a column stripe to arrive adds the stripe's slabs (in slab order) and applies
    LeakyReLU' -- one launch instead of ``vunet_seq_dx`` + ``vunet_seq_dz_finish``.  Whichever workgroup that is, the values are
    those of the two launches: three fused steps are bit-identical, eagerly issued and replayed from the graph (the arrival
    counters are back at zero after every launch)."""
    chan, mid, depth, n_flows, bsz = shape
    res = {}
    for mode in ("two", "one", "one-graph"):
        flow, _ = _random_flow(chan, mid, depth, n_flows, 43)
        eng = flow.flow.train_engine(lr=1e-4, betas=(0.5, 0.9))
        eng.fused_finish = mode != "two"
        eng.graph.enabled = mode == "one-graph"
        logs = [eng.train_step(seeded_randn(f"ff.b{it}", (bsz, chan), 43).cuda(), torch.zeros(bsz, chan, device="cuda")).tolist()
                for it in range(3)]
        res[mode] = (logs, {k: v.detach().clone() for k, v in flow.state_dict().items()})
        assert int(eng._plan(bsz)["dx_cnt"].abs().sum()) == 0
    for mode in ("one", "one-graph"):
        assert res[mode][0] == res["two"][0], mode
        for k, v in res["two"][1].items():
            assert torch.equal(v, res[mode][1][k]), (mode, k)


@pytest.mark.parametrize("shape", [(96, 160, 2, 2, 17), (33, 48, 1, 2, 4), (1024, 2048, 2, 2, 64)])
def test_one_pass_backward_matches_the_two_pass_one(shape):
    """``vunet_seq_dwx`` (the update sweep also forms the layer's input gradient: one pass over W; off by default, it is slower)
    against the default backward pass: the same products in another summation order -- two fused steps agree to rounding, and the
    autograd form hands out the same gradients."""
    chan, mid, depth, n_flows, bsz = shape
    res = {}
    for fused in (True, False):
        flow, _ = _random_flow(chan, mid, depth, n_flows, 43)
        eng = flow.flow.train_engine(lr=1e-4, betas=(0.5, 0.9))
        eng.fused_dx = fused
        eng.graph.enabled = False
        x = seeded_randn("op.x", (bsz, chan), 43).cuda().requires_grad_(True)
        z, logdet = flow(x)
        (z.square().sum() * 0.5 - logdet.sum()).backward()
        grads = {n: p.grad.detach().clone() for n, p in flow.named_parameters()}
        gx = x.grad.detach().clone()
        logs = [eng.train_step(seeded_randn(f"op.b{it}", (bsz, chan), 43).cuda(), torch.zeros(bsz, chan, device="cuda")).tolist()
                for it in range(2)]
        res[fused] = (logs, grads, gx)
    for a, b in zip(res[True][0], res[False][0]):
        assert all(abs(u - v) <= 1e-5 * abs(v) + 1e-6 for u, v in zip(a, b)), (a, b)
    assert _rel(res[True][2], res[False][2]) <= 2e-5
    assert max(_rel(g, res[False][1][n]) for n, g in res[True][1].items()) <= 5e-5


def test_flow_training_refuses_what_it_cannot_do():
    flow, _ = _random_flow(64, 96, 1, 2, 3)
    eng = flow.flow.train_engine(lr=1e-3)
    with pytest.raises(ValueError):
        eng.train_step(torch.randn(65, 64, device="cuda"))
    with pytest.raises(RuntimeError):
        eng.train_step(torch.randn(4, 64))                       # a CPU batch
    z, logdet = flow(torch.randn(4, 64, device="cuda"))
    flow(torch.randn(4, 64, device="cuda"))                      # a second pass replaces the first one's activations
    with pytest.raises(RuntimeError):
        z.sum().backward()
    with pytest.raises(RuntimeError):
        flow.reverse(torch.randn(4, 64, device="cuda"))          # the reverse direction stays inference only


# ---------------------------------------------------------------- the two matrix kernels on their own, through the C ABI
@pytest.mark.parametrize("B,M,K,nets,S", [(64, 2048, 2048, 2, 4), (64, 2048, 512, 2, 16), (64, 512, 2048, 2, 4), (16, 2048, 2048, 2, 4),
                                          (33, 1024, 1088, 1, 2), (48, 256, 64, 1, 1), (5, 4096, 1088, 1, 1), (64, 4096, 1024, 1, 8)])
def test_dx_kernel_vs_float64(B, M, K, nets, S):
    """vunet_seq_dx: dX = dZ . W as S raw slabs (both wave counts of the kernel, every batch-tile count) vs a float64 product."""
    import ctypes
    from behavior_driven_video_synthesis_amd import seq_train as T
    from behavior_driven_video_synthesis_amd.ops import _call, _p, _stream
    g = torch.Generator().manual_seed(B * 7 + M + K)
    bp = (B + 15) // 16 * 16
    w = [torch.randn(M, K, generator=g).cuda() for _ in range(nets)]
    dz = torch.zeros(nets, bp, M)
    dz[:, :B] = torch.randn(nets, B, M, generator=g)
    dz = dz.cuda()
    raw = torch.full((nets, S, bp, K), float("nan"), device="cuda")
    d = T.SeqDxDesc(B, M, K, nets, S, 0)
    _call("vunet_seq_dx", ctypes.byref(d), _p(w[0]), _p(w[1] if nets > 1 else None), _p(dz), _p(raw), _stream())
    got = raw.double().sum(dim=1)
    for n in range(nets):
        want = dz[n].double() @ w[n].double()
        assert _rel(got[n], want) <= 2e-6, (n, _rel(got[n], want))
        # each slab is the product over its own row range
        rows = M // S
        for s in (0, S - 1):
            part = dz[n][:, s * rows:(s + 1) * rows].double() @ w[n][s * rows:(s + 1) * rows].double()
            assert _rel(raw[n, s], part) <= 2e-6


@pytest.mark.parametrize("B", [7, 16, 40, 64])
def test_dw_kernel_write_mode_vs_float64(B):
    """vunet_seq_dw without Adam: the tiles of dW = dZ^T . X and the bias gradient of a two-layer table (different shapes, a
    first layer that reads its input from longer rows and masks its padding columns)."""
    from behavior_driven_video_synthesis_amd import seq_train as T
    from behavior_driven_video_synthesis_amd.ops import _call, _p, _stream
    g = torch.Generator().manual_seed(B)
    bp = (B + 15) // 16 * 16
    shapes = [(128, 192, 256, 150), (2048, 512, 512, 512)]    # M, K, ldx, valid columns
    entries, keep, tile = [], [], 0
    for (m, k, ldx, kv) in shapes:
        dz, x = torch.zeros(bp, m), torch.zeros(bp, ldx)
        dz[:B], x[:B] = torch.randn(B, m, generator=g), torch.randn(B, ldx, generator=g)
        dz, x = dz.cuda(), x.cuda()
        gw, gb = torch.full((m, k), float("nan"), device="cuda"), torch.full((m,), float("nan"), device="cuda")
        bias = torch.zeros(m, device="cuda")
        entries.append(T.SeqDwLayer(None, None, None, gw.data_ptr(), bias.data_ptr(), None, None, gb.data_ptr(), dz.data_ptr(),
                                    x.data_ptr(), m, k, m, ldx, tile, k // 64, kv, 1, 0, 0, None))
        tile += (m // 64) * (k // 64)
        keep.append((dz, x, gw, gb, bias, kv, k))
    tab = T._table(entries, "cuda")
    _call("vunet_seq_dw", _p(tab), len(entries), 0, tile, B, None, _stream())
    for dz, x, gw, gb, _, kv, k in keep:
        want = dz.double().t() @ x[:, :k].double()
        want[:, kv:] = 0
        assert _rel(gw, want) <= 2e-6
        assert _rel(gb, dz.double().sum(0)) <= 2e-6


# ================================================================ first stage: the behaviour cVAE (experiments/behavior_net.py:591-660)
def _cvae_trainer(meta, graph):
    from behavior_driven_video_synthesis_amd.experiments.behavior_net import BehaviorNet, DEFAULT_CONFIG
    import copy
    cfg = copy.deepcopy(DEFAULT_CONFIG)
    cfg["architecture"].update(dim_hidden_b=meta["kw"]["dim_hidden_b"], n_flows=1, flow_mid_channels_factor=1, flow_hidden_depth=1)
    cfg["training"].update(batch_size=meta["batch"], lr_init=meta["lr"], recon_loss_weight=meta["recon_loss_weight"],
                           gamma_init=meta["gamma_init"], gamma_step=meta["gamma_step"], information_max=meta["imax"])
    tr = BehaviorNet(cfg, n_kps=meta["kw"]["n_kps"], hip_graph=graph)
    sd = synth_behavior_state(meta["shapes"], meta["seed"], {})
    tr.net.load_state_dict(sd)
    return tr, sd


def _check_cvae_final(net, meta, arr, tol):
    fin = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    worst = 0.0
    for k, v in arr.items():
        if k.startswith("final."):
            t = fin[k[6:]]
            worst = max(worst, _rel(t[:24] if t.dim() == 2 and t.shape[0] > 64 else t, v))
    assert worst <= tol, f"parameters after the last step: {worst:.2e} of max|.|"
    for k, (s, a) in meta["checksums"].items():
        assert abs(float(fin[k].double().abs().sum()) - a) <= 1e-4 * a + 1e-6, k
    return worst


@pytest.mark.parametrize("graph", [False, True])
def test_fused_cvae_step_follows_the_reference_trajectory(graph):
    """``BehaviorNet.train_fn`` (forward with everything kept, loss, BPTT, fused Adam, gamma controller -- one recorded graph)
    for the fixture's three steps: every step's log and per-frame errors, then the parameters and Adam moments."""
    meta, arr = load_golden("g11_cvae_training")
    seed = meta["seed"]
    tr, _ = _cvae_trainer(meta, graph)
    bsz, t_len, n_kps, hid = meta["batch"], meta["seq_len"], meta["kw"]["n_kps"], meta["kw"]["dim_hidden_b"]
    for it in range(meta["steps"]):
        kps = 0.5 * seeded_randn(f"cvae.kps{it}", (bsz, t_len + 1, n_kps), seed)
        eps = seeded_randn(f"cvae.s{it}.eps0", (bsz, hid), seed).cuda()
        out = tr.train_fn({"keypoints": kps.cuda()}, eps=eps)
        want = meta["logs"][it]
        for k in ("loss", "loss_recon", "kl_loss", "gamma", "mu_s", "logstd_s"):
            assert abs(out[k] - want[k]) <= 2e-5 * abs(want[k]) + 2e-6, (it, k, out[k], want[k])
        assert _rel(out["loss_per_seq_recon"], arr[f"per_seq{it}"]) <= 2e-5
        assert out["seq_len"] == t_len
    worst = _check_cvae_final(tr.net, meta, arr, 3e-5)     # measured 2.3e-6
    st = tr.optimizer.state_dict()
    names = [n for n, _ in tr.net.named_parameters()]
    assert int(st["state"][0]["step"]) == meta["adam_step"] and [g["name"] for g in st["param_groups"]] == ["z_enc", "dec"]
    for k, v in arr.items():
        if k.startswith("exp_avg."):
            e = st["state"][names.index(k[8:])]["exp_avg"]
            assert _rel(e[:24] if e.dim() == 2 and e.shape[0] > 64 else e, v) <= 3e-5, k
    print(f"\n[g11 graph={graph}] parameters after 3 fused steps: {worst:.2e} of max|.|")


def test_cvae_autograd_with_torch_adam_follows_the_reference_trajectory():
    """What an unchanged ``train_fn`` does through the drop-in: ``net(seq_b, seq_b, seq_len)`` under autograd, the losses in
    torch, ``backward()`` into the HIP back-propagation through time, ``torch.optim.Adam.step()``."""
    from behavior_driven_video_synthesis_amd.models.pose_behavior_rnn import ResidualBehaviorNet
    meta, arr = load_golden("g11_cvae_training")
    seed = meta["seed"]
    net = ResidualBehaviorNet(**meta["kw"])
    net.load_state_dict(synth_behavior_state(meta["shapes"], seed, {}))
    net = net.cuda().train()
    opt = torch.optim.Adam([{"params": net.b_enc.parameters(), "name": "z_enc"}, {"params": net.decoder.parameters(), "name": "dec"}],
                           lr=meta["lr"])
    bsz, t_len, n_kps, hid = meta["batch"], meta["seq_len"], meta["kw"]["n_kps"], meta["kw"]["dim_hidden_b"]
    gamma = meta["gamma_init"]
    for it in range(meta["steps"]):
        kps = (0.5 * seeded_randn(f"cvae.kps{it}", (bsz, t_len + 1, n_kps), seed)).cuda()
        seq_b, target = kps[:, :-1], kps[:, 1:]
        xs, cs, _, bs, mu, logstd, pre = net(seq_b, seq_b, t_len, eps=seeded_randn(f"cvae.s{it}.eps0", (bsz, hid), seed).cuda())
        recon = torch.mean(torch.nn.functional.mse_loss(xs, target, reduction="none"))
        std = torch.exp(logstd)
        kl = (torch.sum(-logstd + 0.5 * (std ** 2 + mu ** 2), dim=-1) - 0.5 * hid).mean()      # lib/losses.py:283-291
        loss = meta["recon_loss_weight"] * recon + gamma * kl
        opt.zero_grad()
        loss.backward()
        opt.step()
        gamma = max(gamma - meta["gamma_step"] * (meta["imax"] - float(kl.detach())), 0)
        want = meta["logs"][it]
        assert abs(float(loss.detach()) - want["loss"]) <= 2e-5 * abs(want["loss"]), (it, float(loss.detach()), want["loss"])
        if it == 0:
            assert _rel(xs, arr["xs0"]) <= 2e-5 and _rel(bs, arr["bs0"]) <= 2e-5
    _check_cvae_final(net, meta, arr, 3e-5)


@pytest.mark.parametrize("bsz,hid,t_in,length,start", [(1, 64, 3, 4, 0), (17, 128, 5, 7, 2), (64, 64, 4, 3, 3)])
def test_cvae_gradients_vs_oracle(bsz, hid, t_in, length, start):
    """Every output of ``net(x1, x2, len, start_frame)`` weighted into a loss -- xs, cs, b, mu, logstd, pre -- and every parameter's
    gradient vs torch.autograd over the oracle (different sequences for the encoder and the start pose, a roll-out longer than
    the input)."""
    from behavior_driven_video_synthesis_amd.models.pose_behavior_rnn import ResidualBehaviorNet
    from oracle import behavior_oracle as B
    n_kps = 51
    net = ResidualBehaviorNet(n_kps, information_bottleneck=True, decoder_arch="lstm", dim_hidden_b=hid)
    sd = synth_behavior_state({k: list(v.shape) for k, v in net.state_dict().items()}, 9, {})
    net.load_state_dict(sd)
    net = net.cuda().train()
    x1 = 0.5 * seeded_randn("cg.x1", (bsz, t_in, n_kps), 9)
    x2 = 0.5 * seeded_randn("cg.x2", (bsz, t_in + 1, n_kps), 9)
    eps = seeded_randn("cg.eps", (bsz, hid), 9)
    ws = {k: seeded_randn(f"cg.w.{k}", shp, 9) for k, shp in dict(xs=(bsz, length, n_kps), cs=(bsz, length, n_kps), b=(bsz, hid),
                                                                    mu=(bsz, hid), logstd=(bsz, hid), pre=(bsz, hid)).items()}
    xs, cs, _, b, mu, logstd, pre = net(x1.cuda(), x2.cuda(), length, start_frame=start, eps=eps.cuda())
    loss = sum((t * ws[k].cuda()).sum() for k, t in dict(xs=xs, cs=cs, b=b, mu=mu, logstd=logstd, pre=pre).items())
    loss.backward()
    ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    rxs, rcs, rb, rmu, rls, rpre = B.behavior_net_forward(ref, x1, x2, length, start_frame=start, eps=eps)
    sum((t * ws[k]).sum() for k, t in dict(xs=rxs, cs=rcs, b=rb, mu=rmu, logstd=rls, pre=rpre).items()).backward()
    assert _rel(xs, rxs) <= 2e-5 and _rel(b, rb) <= 2e-5 and _rel(pre, rpre) <= 2e-5
    worst = 0.0
    for n, q in net.named_parameters():
        assert q.grad is not None, n
        worst = max(worst, _rel(q.grad, ref[n].grad))
    assert worst <= 5e-5, f"{worst:.2e}"


@pytest.mark.parametrize("bsz,hid,t_in,length", [(48, 64, 5, 7), (64, 128, 6, 6), (33, 128, 4, 5), (16, 64, 4, 4)])
def test_h_handed_on_as_tiles_changes_nothing(bsz, hid, t_in, length):
    """``vunet_seq_lstm_gates_tiled_h``: batches of more than 32 rows read the h part of an LSTM step's operand from the tile-major
    copy the previous step left (the x part and the first step's h from the rows).  Outputs and every parameter's gradient are
    bit-identical with the path that reads the rows (33 rows -> 48 padded: tiles; 16 rows: the rows either way)."""
    from behavior_driven_video_synthesis_amd.models.pose_behavior_rnn import ResidualBehaviorNet
    n_kps = 51
    res = {}
    for tiled in (True, False):
        net = ResidualBehaviorNet(n_kps, information_bottleneck=True, decoder_arch="lstm", dim_hidden_b=hid)
        net.load_state_dict(synth_behavior_state({k: list(v.shape) for k, v in net.state_dict().items()}, 17, {}))
        net = net.cuda().train()
        eng = net.train_engine()
        eng.tile_h = tiled
        x1 = (0.5 * seeded_randn("th.x1", (bsz, t_in, n_kps), 17)).cuda()
        x2 = (0.5 * seeded_randn("th.x2", (bsz, t_in + 1, n_kps), 17)).cuda()
        eps = seeded_randn("th.eps", (bsz, hid), 17).cuda()
        xs, cs, _, b, mu, logstd, pre = net(x1, x2, length, start_frame=1, eps=eps)
        ((xs * xs).sum() + (cs * 0.3).sum() + (mu * logstd).sum() + pre.sum()).backward()
        plan = next(v for k, v in eng._plans.items() if k and k[0] == "train")
        assert (plan["ht"] is not None) == (tiled and bsz > 32 and hid % 32 == 0)
        res[tiled] = ([t.detach().clone() for t in (xs, cs, b, mu, logstd, pre)], {n: q.grad.clone() for n, q in net.named_parameters()})
    for a, b_ in zip(res[True][0], res[False][0]):
        assert torch.equal(a, b_)
    for n, g in res[True][1].items():
        assert torch.equal(g, res[False][1][n]), n


def test_cvae_step_at_the_reference_size_vs_oracle():
    """config/behavior_net.yaml: dim_hidden_b 1024, 51 pose dimensions, batch 64, 50 frames: one fused step (graph replay equals
    eager issue bit for bit) vs the oracle's step: the step's log, every parameter's first moment (= 0.1 x its gradient)."""
    from behavior_driven_video_synthesis_amd.experiments.behavior_net import BehaviorNet, DEFAULT_CONFIG
    from oracle import behavior_oracle as B
    import copy
    cfg = copy.deepcopy(DEFAULT_CONFIG)
    cfg["architecture"].update(n_flows=1, flow_mid_channels_factor=1, flow_hidden_depth=1)
    cfg["training"].update(gamma_init=0.01, gamma_step=1e-3, information_max=20)
    kps = [0.5 * seeded_randn(f"cw.kps{i}", (64, 51, 51), 13) for i in range(2)]
    eps = [seeded_randn(f"cw.eps{i}", (64, 1024), 13) for i in range(2)]
    runs = {}
    for graph in (False, True):
        tr = BehaviorNet(cfg, n_kps=51, hip_graph=graph)
        sd = synth_behavior_state({k: list(v.shape) for k, v in tr.net.state_dict().items()}, 13, {})
        sd["decoder.n_out.weight"] *= 0.05
        tr.net.load_state_dict(sd)
        outs = [tr.train_fn({"keypoints": kps[i].cuda()}, eps=eps[i].cuda()) for i in range(2)]
        runs[graph] = (outs, {k: v.detach().clone() for k, v in tr.net.state_dict().items()}, tr)
    for k, v in runs[False][1].items():
        assert torch.equal(v, runs[True][1][k]), k
    assert all(runs[False][0][i]["loss"] == runs[True][0][i]["loss"] for i in range(2))
    ref = {k: v.clone() for k, v in sd.items()}
    opt = B.behavior_optimizer(ref, cfg["training"]["lr_init"])
    log, gamma, _ = B.cvae_train_step(ref, opt, kps[0], eps[0], 0.01, 2.5, 1e-3, 20.0)
    got = runs[False][0][0]
    for k in ("loss", "loss_recon", "kl_loss", "gamma"):
        assert abs(got[k] - log[k]) <= 2e-5 * abs(log[k]) + 1e-7, (k, got[k], log[k])
    # one-step moments: run a fresh trainer for exactly one step
    tr = BehaviorNet(cfg, n_kps=51, hip_graph=False)
    tr.net.load_state_dict(sd)
    tr.train_fn({"keypoints": kps[0].cuda()}, eps=eps[0].cuda())
    mine, theirs = tr.optimizer.state_dict()["state"], opt.state_dict()["state"]
    enc, dec = B.behavior_parameters(ref)
    order = [n for n, _ in tr.net.named_parameters()]
    worst = 0.0
    for i, n in enumerate(enc + dec):
        worst = max(worst, _rel(mine[order.index(n)]["exp_avg"], theirs[i]["exp_avg"]))
    print(f"\n[cVAE 1024 / 51 / 64 x 50] first moments after one step: {worst:.2e} of max|.|")
    assert worst <= 3e-5      # measured 2.4e-6


# ================================================================ the trainer's surface (experiments/behavior_net.py)
def _small_trainer(only_flow, graph=True, seed=23):
    from behavior_driven_video_synthesis_amd.experiments.behavior_net import BehaviorNet, DEFAULT_CONFIG
    import copy
    cfg = copy.deepcopy(DEFAULT_CONFIG)
    cfg["architecture"].update(dim_hidden_b=64, n_flows=2, flow_mid_channels_factor=2, flow_hidden_depth=1)
    cfg["training"].update(batch_size=6, lr_init=1e-3, gamma_init=0.05, gamma_step=1e-3, information_max=5, flow_lr=5e-5, weight_decay=1e-3,
                           only_flow=only_flow)
    tr = BehaviorNet(cfg, n_kps=51, hip_graph=graph)
    fsd = synth_behavior_state({k: list(v.shape) for k, v in tr.latent_flow.state_dict().items()}, seed,
                               {k: v.cpu() for k, v in tr.latent_flow.state_dict().items() if k.endswith("_shuffle_idx")})
    nsd = synth_behavior_state({k: list(v.shape) for k, v in tr.net.state_dict().items()}, seed, {})
    tr.latent_flow.load_state_dict(fsd)
    tr.net.load_state_dict(nsd)
    return tr, fsd, nsd


def test_flow_stage_through_train_fn_vs_oracle():
    """``only_flow`` (experiments/behavior_net.py:591-604, :703-714): the net encodes under no_grad -- its parameters do not move --,
    the flow takes one optimisation step on ``bs.detach()``; the log carries the flow's four scalars and the net's recon / KL.
    Three steps (eager, recorded, replayed) against the oracle driven the same way."""
    from oracle import behavior_oracle as B
    tr, fsd, nsd = _small_trainer(True)
    fref = {k: v.clone() for k, v in fsd.items()}
    opt = B.flow_optimizer(fref, 5e-5 * 6, 1e-3)
    for it in range(3):
        kps = 0.5 * seeded_randn(f"fs.kps{it}", (6, 8, 51), 23)
        eps = seeded_randn(f"fs.eps{it}", (6, 64), 23)
        out = tr.train_fn({"keypoints": kps.cuda()}, eps=eps.cuda(), noise=torch.zeros(6, 64, device="cuda"))
        with torch.no_grad():
            xs, cs, b, mu, logstd, pre = B.behavior_net_forward(nsd, kps[:, :-1], kps[:, :-1], 7, 0, eps=eps)
            recon = torch.nn.functional.mse_loss(xs, kps[:, 1:])
            kl = B.kl_loss(mu, logstd)
        log = B.flow_train_step(fref, opt, b)
        for k, want in (("flow_loss", log["flow_loss"]), ("nll_loss", log["nll_loss"]), ("nlogdet_loss", log["nlogdet_loss"]),
                        ("loss_recon", float(recon)), ("kl_loss", float(kl)), ("mu_s", float(mu.mean())), ("logstd_s", float(logstd.mean()))):
            assert abs(out[k] - want) <= 5e-5 * abs(want) + 2e-6, (it, k, out[k], want)
        assert out["seq_len"] == 7 and out["gamma"] == pytest.approx(0.05)      # the controller does not run in this stage
    for k, v in tr.net.state_dict().items():
        assert torch.equal(v.cpu(), nsd[k]), k
    # (Adam moves an element by about lr whatever its gradient's size: one whose gradient is summation noise may go the other way)
    lr, far, total, worst = 5e-5 * 6, 0, 0, 0.0
    for k, v in tr.latent_flow.state_dict().items():
        if v.dtype.is_floating_point and v.dim() > 0:
            d = (v.detach().cpu().double() - fref[k].detach().double()).abs()
            worst, far, total = max(worst, float(d.max())), far + int((d > 0.02 * lr).sum()), total + d.numel()
    assert worst <= 2.001 * 3 * lr and far <= max(3, 1e-4 * total), (worst / lr, far, total)


@pytest.mark.parametrize("only_flow", [False, True])
def test_trainer_checkpoint_round_trip(only_flow):
    """``state_dict`` / ``load_state_dict`` (model, optimizer, flow, flow optimizer in torch.optim.Adam's layout, gamma): a trainer
    restored from a checkpoint takes the next step bit for bit like the one that wrote it."""
    tr, _, _ = _small_trainer(only_flow, graph=False)
    batches = [({"keypoints": (0.5 * seeded_randn(f"ck.kps{it}", (6, 8, 51), 29)).cuda()}, seeded_randn(f"ck.eps{it}", (6, 64), 29).cuda())
               for it in range(3)]
    noise = torch.zeros(6, 64, device="cuda")
    for b_, e_ in batches[:2]:
        tr.train_fn(b_, eps=e_, noise=noise)
    import copy
    ck = copy.deepcopy(tr.state_dict())     # (what torch.save would freeze: state_dict() hands out the live tensors, as nn.Module's does)
    assert set(ck) == {"model", "optimizer", "flow", "flow_optimizer", "gamma"}
    torch.optim.Adam(tr.latent_flow.parameters()).load_state_dict(ck["flow_optimizer"])      # (it is torch's layout)
    out_a = tr.train_fn(batches[2][0], eps=batches[2][1], noise=noise)
    tr2, _, _ = _small_trainer(only_flow, graph=False, seed=77)        # different weights: everything must come from the checkpoint
    tr2.load_state_dict(ck)
    out_b = tr2.train_fn(batches[2][0], eps=batches[2][1], noise=noise)
    for k, v in out_a.items():
        assert np.array_equal(np.asarray(v), np.asarray(out_b[k])), k
    for k, v in tr.net.state_dict().items():
        assert torch.equal(v, tr2.net.state_dict()[k]), k
    for k, v in tr.latent_flow.state_dict().items():
        assert torch.equal(v, tr2.latent_flow.state_dict()[k]), k


def test_learning_rate_schedule_reaches_the_recorded_step():
    """``MultiStepLR`` steps the net's learning rate between epochs (experiments/behavior_net.py:337-339): ``set_lr`` must take
    effect on the RECORDED step (the rate lives in device memory, nothing is re-recorded)."""
    tr, _, nsd = _small_trainer(False, graph=True)
    batch = {"keypoints": (0.5 * seeded_randn("lr.kps", (6, 8, 51), 31)).cuda()}
    eps = seeded_randn("lr.eps", (6, 64), 31).cuda()
    for _ in range(3):
        tr.train_fn(batch, eps=eps)          # eager, recorded, replayed
    before = {k: v.detach().clone() for k, v in tr.net.state_dict().items()}
    tr.set_lr(0.0)
    tr.train_fn(batch, eps=eps)
    assert all(torch.equal(v, before[k]) for k, v in tr.net.state_dict().items())       # a step of rate 0 moves nothing
    tr.set_lr(1e-3)
    tr.train_fn(batch, eps=eps)
    assert any(not torch.equal(v, before[k]) for k, v in tr.net.state_dict().items())
    assert tr.optimizer.state_dict()["state"][0]["step"] == 5
