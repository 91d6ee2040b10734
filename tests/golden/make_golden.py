#!/usr/bin/env python3
"""Golden-vector generator.  Runs ONLY in the build container (needs /root/reference).

Imports the upstream reference's own modules (lib/modules.py, models/vunets.py and -- behind
throw-away stub modules for the absent third-party packages -- lib/losses.py,
models/synth_discriminator.py, models/imagenet_pretrained.py), drives them on seeded synthetic
inputs / parameters (tests/golden/synth.py) and writes small ``.npz`` fixtures with the expected
outputs.  The fixtures hold data only (shapes, seeds, inputs' recipe, expected outputs).

    python tests/golden/make_golden.py            # regenerate tests/golden/*.npz

The tests never import the reference; they rebuild inputs from the recipe and compare the oracle
(oracle/vunet_oracle.py) against the stored outputs.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
REF = os.environ.get("VUNET_REFERENCE", "/root/reference")

from synth import seeded_randn, synth_behavior_state, synth_image, synth_state_dict  # noqa: E402


def _install_stubs():
    """Absent third-party packages that the reference imports at module import time only."""
    class _Any(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return type(k, (), {"__init__": lambda self, *a, **kw: None})

    for name in ["cv2", "kornia", "ignite", "ignite.engine", "ignite.handlers", "ignite.metrics",
                 "ignite.contrib", "ignite.contrib.handlers", "torch.utils.tensorboard", "torchvision",
                 "torchvision.utils", "torchvision.models", "torchvision.transforms", "wandb",
                 "tqdm.autonotebook", "matplotlib.backends.backend_agg"]:
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = _Any(name)


_install_stubs()
sys.path.insert(0, REF)
from lib import modules as rm  # noqa: E402
from models import vunets as rv  # noqa: E402
from lib import losses as rl  # noqa: E402
from models import synth_discriminator as rd  # noqa: E402
from models import imagenet_pretrained as rp  # noqa: E402


def shapes_of(mod):
    return {k: list(v.shape) for k, v in mod.state_dict().items()}


def load_synth(mod, seed):
    sh = shapes_of(mod)
    mod.load_state_dict(synth_state_dict(sh, seed))
    return sh


def npify(d):
    return {k: v.detach().cpu().numpy() for k, v in d.items()}


def save(name, meta, arrays):
    arrays = dict(arrays)
    arrays["__meta__"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


class FixedNoise:
    """Replace torch.randn_like by a recorded, seeded sequence (SURVEY 8c determinism recipe)."""

    def __init__(self, tag, seed):
        self.tag, self.seed, self.i, self.shapes = tag, seed, 0, []

    def __enter__(self):
        self._orig = torch.randn_like

        def fake(t, **kw):
            e = seeded_randn(f"{self.tag}.eps{self.i}", tuple(t.shape), self.seed)
            self.shapes.append(list(t.shape))
            self.i += 1
            return e
        torch.randn_like = fake
        return self

    def __exit__(self, *a):
        torch.randn_like = self._orig


# ---------------------------------------------------------------- G1 primitives
def g1_primitives():
    seed = 11
    arrays, meta = {}, {"seed": seed, "cases": {}}

    def run(case, mod, inputs, call):
        sh = load_synth(mod, seed) if len(list(mod.parameters())) else {}
        mod.train()
        ins = [t.clone().requires_grad_(True) for t in inputs]
        y = call(mod, *ins)
        wgt = seeded_randn(case + ".wgt", tuple(y.shape), seed)
        (y * wgt).sum().backward()
        arrays[case + ".y"] = y.detach().numpy()
        for i, t in enumerate(ins):
            arrays[f"{case}.gin{i}"] = t.grad.numpy()
        for k, p in mod.named_parameters():
            arrays[f"{case}.gp.{k}"] = p.grad.numpy()
        meta["cases"][case] = {"shapes": sh, "inputs": [list(t.shape) for t in inputs], "out": list(y.shape)}

    x = lambda case, *s: synth_image(case + ".x", s, seed)  # noqa: E731
    a = lambda case, *s: synth_image(case + ".a", s, seed)  # noqa: E731
    run("nc_k3s1", rm.NormConv2d(6, 10, 3, 1, 1), [x("nc_k3s1", 2, 6, 12, 12)], lambda m, t: m(t))
    run("nc_k1", rm.NormConv2d(3, 8, 1), [x("nc_k1", 2, 3, 16, 16)], lambda m, t: m(t))
    run("nc_k3valid", rm.NormConv2d(3, 16, 3), [x("nc_k3valid", 2, 3, 10, 10)], lambda m, t: m(t))
    run("down", rm.Downsample(8, 16), [x("down", 2, 8, 16, 16)], lambda m, t: m(t))
    run("up", rm.Upsample(8, 4), [x("up", 2, 8, 8, 8)], lambda m, t: m(t))
    run("rnb_plain", rm.VunetRNB(8), [x("rnb_plain", 2, 8, 12, 12)], lambda m, t: m(t))
    run("rnb_res", rm.VunetRNB(8, a_channels=8, residual=True),
        [x("rnb_res", 2, 8, 8, 8), a("rnb_res", 2, 8, 8, 8)], lambda m, t, u: m(t, u))
    run("rnb_res2", rm.VunetRNB(8, a_channels=16, residual=True),
        [x("rnb_res2", 2, 8, 4, 4), a("rnb_res2", 2, 16, 4, 4)], lambda m, t, u: m(t, u))
    run("s2d", rm.SpaceToDepth(2), [x("s2d", 2, 3, 8, 12)], lambda m, t: m(t))
    run("d2s", rm.DepthToSpace(2), [x("d2s", 2, 12, 4, 6)], lambda m, t: m(t))
    run("l2nc", rm.L2NormConv2d(6, 8, 3, 1, 1, bias=False), [x("l2nc", 2, 6, 8, 8)], lambda m, t: m(t))
    run("lnc", rm.LayerNormConv2d(6, 8, 3, 1, 1), [x("lnc", 2, 6, 8, 8)], lambda m, t: m(t))
    save("g1_primitives", meta, arrays)


# ---------------------------------------------------------------- G2 whole models
ALTER_CFG = dict(spatial_size=32, bottleneck_factor=2, box_factor=2, n_scales=0, n_latent_scales=2,
                 conv_layer_type="l1", nf_start=8, nf_max=16, subpixel_upsampling=True, dropout_prob=0.0,
                 # unrelated keys that the reference splats in and must be ignored
                 dataset="Human3.6m", n_rnb=2, cvae=False, linear_width_factor=1)
ALTER_BOX_CFG = dict(ALTER_CFG, bottleneck_factor=1, box_factor=1)
ORG_CFG = dict(ALTER_CFG, nf_start=4, nf_max=8)


def model_loss(outs, tag, seed):
    tot = 0.0
    flat = []
    for o in outs:
        if isinstance(o, (list, tuple)):
            flat.extend(o)
        else:
            flat.append(o)
    for i, o in enumerate(flat):
        tot = tot + (o * seeded_randn(f"{tag}.lw{i}", tuple(o.shape), seed)).sum()
    return tot, flat


def g2_models():
    seed = 21
    # ---- VunetAlter, 3-channel appearance input
    for tag, cfg, ncx, xshape in [("alter", ALTER_CFG, 3, (2, 3, 32, 32)),
                                  ("alter_box", ALTER_BOX_CFG, 6, (2, 6, 16, 16))]:
        arrays = {}
        net = rv.VunetAlter(n_channels_x=ncx, **cfg)
        sh = load_synth(net, seed)
        net.train()
        x = synth_image(tag + ".x", xshape, seed).requires_grad_(True)
        c = synth_image(tag + ".c", (2, 3, 32, 32), seed).requires_grad_(True)
        with FixedNoise(tag, seed) as fn:
            img, means, logstds, acts = net(x, c)
        loss, flat = model_loss([img, means, logstds], tag, seed)
        loss.backward()
        arrays["img"] = img.detach().numpy()
        for i, (m, l) in enumerate(zip(means, logstds)):
            arrays[f"mean{i}"] = m.detach().numpy()
            arrays[f"logstd{i}"] = l.detach().numpy()
        arrays["gx"] = x.grad.numpy()
        arrays["gc"] = c.grad.numpy()
        gsum = {}
        for k, p in net.named_parameters():
            # parameters behind the discarded ``es`` output of ``ed`` get no gradient (None)
            gsum[k] = None if p.grad is None else [float(p.grad.double().sum()), float(p.grad.double().abs().sum())]
        for k in ["eu.nin.conv.weight_v", "eu.blocks.0.conv.gamma", "ed.make_logstds.1.conv.weight_g",
                  "ed.blocks.1.nin.conv.weight_v", "du.downs.0.down.conv.bias", "dd.ups.0.up.conv.weight_v",
                  "dd.auto_blocks.1.conv.beta", "dd.out_conv.conv.weight_v"]:
            arrays["gp." + k] = dict(net.named_parameters())[k].grad.numpy()
        with FixedNoise(tag + ".tr", seed):
            net.eval()
            with torch.no_grad():
                arrays["transfer"] = net.transfer(x.detach(), c.detach()).numpy()
        with FixedNoise(tag + ".tf", seed) as fn2:
            with torch.no_grad():
                arrays["test_forward"] = net.test_forward(c.detach()).numpy()
        meta = {"seed": seed, "cfg": cfg, "n_channels_x": ncx, "shapes": sh, "x": list(xshape),
                "c": [2, 3, 32, 32], "eps_shapes": fn.shapes, "tf_eps_shapes": fn2.shapes,
                "grad_sums": gsum, "n_params": sum(p.numel() for p in net.parameters())}
        save("g2_" + tag, meta, arrays)

    # ---- VunetOrg
    tag, arrays = "org", {}
    net = rv.VunetOrg(n_channels_x=3, **ORG_CFG)
    sh = load_synth(net, seed)
    net.train()
    x = synth_image(tag + ".x", (2, 3, 32, 32), seed).requires_grad_(True)
    c = synth_image(tag + ".c", (2, 3, 32, 32), seed).requires_grad_(True)
    with FixedNoise(tag, seed) as fn:
        img, qs, ps, acts = net(x, c)
    loss, flat = model_loss([img, qs, ps], tag, seed)
    loss.backward()
    arrays["img"] = img.detach().numpy()
    for i, (q, p_) in enumerate(zip(qs, ps)):
        arrays[f"q{i}"] = q.detach().numpy()
        arrays[f"p{i}"] = p_.detach().numpy()
    arrays["gx"] = x.grad.numpy()
    arrays["gc"] = c.grad.numpy()
    arrays["kl"] = rl.compute_kl_loss(ps, qs).detach().numpy()
    gsum = {k: (None if p.grad is None else [float(p.grad.double().sum()), float(p.grad.double().abs().sum())])
            for k, p in net.named_parameters()}
    meta = {"seed": seed, "cfg": ORG_CFG, "n_channels_x": 3, "shapes": sh, "x": [2, 3, 32, 32],
            "c": [2, 3, 32, 32], "eps_shapes": fn.shapes, "grad_sums": gsum}
    save("g2_org", meta, arrays)

    # ---- Regressor (models/vunets.py:786-824)
    reg = rv.Regressor(n_out=34, n_latent_scales=2, nf_max=16, latent_widths=[8, 4], linear_width_factor=1)
    sh = load_synth(reg, seed)
    e0 = seeded_randn("reg.e0", (2, 16, 4, 4), seed)
    e1 = seeded_randn("reg.e1", (2, 16, 8, 8), seed)
    out = reg([e0, e1])
    save("g2_regressor", {"seed": seed, "shapes": sh}, {"out": out.detach().numpy()})


# ---------------------------------------------------------------- G3 losses / VGG
def build_vgg_features(vgg_sd):
    from torch import nn
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
    layers, cin, idx = [], 3, 0
    for v in cfg:
        if v == "M":
            layers.append(nn.MaxPool2d(2, 2))
            idx += 1
        else:
            cout = vgg_sd[f"features.{idx}.weight"].shape[0]
            layers += [nn.Conv2d(cin, cout, 3, padding=1), nn.ReLU(inplace=True)]
            cin = cout
            idx += 2
    feats = nn.Sequential(*layers)
    feats.load_state_dict({k[len("features."):]: v for k, v in vgg_sd.items()})

    class V(nn.Module):
        def __init__(self):
            super().__init__()
            self.features = feats
    return V()


def g3_losses():
    sys.path.insert(0, ROOT)
    from oracle.vunet_oracle import make_synthetic_vgg19
    seed = 31
    arrays = {}
    means = [seeded_randn("kl.m0", (3, 16, 4, 4), seed), seeded_randn("kl.m1", (3, 16, 8, 8), seed)]
    logstds = [torch.sigmoid(seeded_randn("kl.l0", (3, 16, 4, 4), seed)),
               torch.sigmoid(seeded_randn("kl.l1", (3, 16, 8, 8), seed))]
    arrays["kl"] = rl.compute_kl_with_prior(means, logstds).numpy()
    arrays["latent_kl"] = rl.compute_kl_loss([means[0]], [logstds[0]]).numpy()

    # full-width synthetic VGG19 on a 32x32 input: taps of the reference's PerceptualVGG
    vsd = make_synthetic_vgg19(seed=1234)
    pv = rp.PerceptualVGG(build_vgg_features(vsd), [1.0, 0.5, 2.0, 1.5, 0.25, 3.0])
    pv.eval()
    t = synth_image("vgg.t", (2, 3, 32, 32), seed)
    p = synth_image("vgg.p", (2, 3, 32, 32), seed).requires_grad_(True)
    feats = pv(t)
    for k, v in feats.items():
        v = v.detach()
        arrays["tap." + k + ".stats"] = np.array([v.double().mean(), v.double().abs().mean(), v.double().std()])
        arrays["tap." + k + ".head"] = v.flatten()[:64].numpy()
    ld = rl.vgg_loss(pv, t, p)
    tot = torch.stack([ld[k] for k in ld], 0).sum()
    tot.backward()
    for k, v in ld.items():
        arrays["vggloss." + k] = v.detach().numpy()
    arrays["vggloss.gp"] = p.grad.numpy()
    save("g3_losses", {"seed": seed, "vgg_seed": 1234, "loss_weights": [1.0, 0.5, 2.0, 1.5, 0.25, 3.0],
                       "tap_order": list(feats.keys())}, arrays)


# ---------------------------------------------------------------- G4 discriminators
def g4_discriminators():
    seed = 41
    arrays, meta = {}, {"seed": seed}
    pd = rd.PartDiscriminator(n_scales=2, part_size=16)
    meta["part_shapes"] = load_synth(pd, seed)
    x = synth_image("pd.x", (2, 3, 18, 18), seed).requires_grad_(True)
    out = pd(x)
    reg = rd.compute_grad2(out, x).mean()
    (out.sum() + 10.0 * reg).backward()
    arrays["pd.out"] = out.detach().numpy()
    arrays["pd.reg"] = reg.detach().numpy()
    arrays["pd.gx"] = x.grad.numpy()
    meta["pd_grad_sums"] = {k: [float(p.grad.double().sum()), float(p.grad.double().abs().sum())]
                            for k, p in pd.named_parameters()}

    pg = rd.PatchGANDiscriminator(3, ndf=8, n_layers=3)
    meta["patch_shapes"] = load_synth(pg, seed)
    x = synth_image("pg.x", (2, 3, 32, 32), seed).requires_grad_(True)
    out = pg(x)
    w = seeded_randn("pg.w", tuple(out.shape), seed)
    (out * w).sum().backward()
    arrays["pg.out"] = out.detach().numpy()
    arrays["pg.gx"] = x.grad.numpy()
    meta["pg_grad_sums"] = {k: [float(p.grad.double().sum()), float(p.grad.double().abs().sum())]
                            for k, p in pg.named_parameters()}

    # DiscTrainer: one D step + generator loss (models/synth_discriminator.py:139-206)
    class Gen(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.last = torch.nn.Conv2d(3, 3, 1)
    gen = Gen()
    load_synth(gen, seed)
    tr = rd.DiscTrainer(gen, {"pd_scales": 2, "adam_beta": (0.5, 0.9), "save_intervall": 10},
                        spatial_size=64, grad_pen=True, lambda_gp=10, grad_weighting=True)
    load_synth(tr.disc, seed)
    tr.init_training([torch.device("cpu")], lr=1e-3)
    real = synth_image("dt.real", (2, 3, 18, 18), seed)
    zin = synth_image("dt.z", (2, 3, 18, 18), seed)
    fake = gen.last(zin)
    dout = tr.train_disc(real, fake.detach())
    meta["train_disc"] = dout
    meta["disc_after"] = {k: [float(v.double().sum()), float(v.double().abs().sum())]
                          for k, v in tr.disc.state_dict().items()}
    fake = gen.last(zin)
    pre = (fake - real).abs().mean()
    gl, wgt = tr.get_genloss(fake, pre, gen.last.weight)
    arrays["dt.gen_loss"] = gl.detach().numpy()
    arrays["dt.loss_weight"] = wgt.detach().numpy()
    save("g4_discriminators", meta, arrays)


# ---------------------------------------------------------------- G5 K-step training trajectory
def g5_trajectory():
    from oracle.vunet_oracle import make_synthetic_vgg19
    seed, K = 51, 3
    cfg = dict(ALTER_CFG)
    net = rv.VunetAlter(n_channels_x=3, **cfg)
    sh = load_synth(net, seed)
    vsd = make_synthetic_vgg19(seed=77, width_div=8)
    pv = rp.PerceptualVGG(build_vgg_features(vsd), [1.0] * 6)
    pv.eval()
    lr0, betas, gamma_step, imax, n_init, total_steps = 5e-4, (0.5, 0.9), 1e-5, 1000.0, 1, 100
    opt = torch.optim.Adam([{"params": getattr(net, n).parameters(), "name": n} for n in ["eu", "ed", "du", "dd"]],
                           lr=lr0, betas=betas)
    gamma, lr = 0.5, lr0
    rec = []
    net.train()
    for it in range(1, K + 1):
        x = synth_image(f"traj.x{it}", (2, 3, 32, 32), seed)
        c = synth_image(f"traj.c{it}", (2, 3, 32, 32), seed)
        with FixedNoise(f"traj.{it}", seed):
            img, means, logstds, _ = net(x, c)
        ld = rl.vgg_loss(pv, x, img)
        ll = 1.0 * torch.sum(torch.stack([ld[k] for k in ld], dim=0))
        kl = rl.compute_kl_with_prior(means, logstds)
        loss = ll
        if it > n_init:
            loss = loss + torch.tensor(gamma) * kl
        opt.zero_grad()
        loss.backward()
        opt.step()
        gamma = max(gamma - gamma_step * (imax - float(kl)), 0)
        rec.append({"it": it, "loss": float(loss), "ll": float(ll), "kl": float(kl), "gamma_after": gamma, "lr": lr})
        lr = float(np.clip(float(0 - lr0) / (total_steps - 0) * (it - 0) + lr0, 0, lr0))
        for pg_ in opt.param_groups:
            pg_["lr"] = lr
    csum = {k: [float(v.double().sum()), float(v.double().abs().sum())] for k, v in net.state_dict().items()}
    meta = {"seed": seed, "cfg": cfg, "shapes": sh, "vgg_seed": 77, "vgg_width_div": 8, "K": K, "lr0": lr0,
            "betas": list(betas), "gamma0": 0.5, "gamma_step": gamma_step, "imax": imax, "n_init_batches": n_init,
            "total_steps": total_steps, "steps": rec, "param_checksums": csum}
    save("g5_trajectory", meta, {"final.dd.out_conv.conv.weight_v": net.state_dict()["dd.out_conv.conv.weight_v"].numpy()})


# ---------------------------------------------------------------- G5c trajectory of the VunetOrg loop (experiments/vunet.py)
def g5_org_trajectory():
    """experiments/vunet.py:248-338,362-371 driven by hand: VunetOrg forward (posterior + autoregressive prior draws) ->
    ll_weight * sum(vgg_loss) + kl_weight * compute_kl_loss(p_means, q_means) -> torch.optim.Adam over the four param groups;
    after every iteration lr <- linear decay to 0 over total_steps and kl_weight <- linear ramp kl_init .. kl_max between
    total_steps // 2 and 3 * total_steps // 4 (so the weight a step uses is the one set after the previous iteration).
    total_steps = 8, K = 7: the ramp (iterations 6 and 7 run with 0.5 and 1.0) is inside the trajectory."""
    from functools import partial
    from lib import utils as ru   # linear_var (lib/utils.py:520-527), the schedule function the reference loop binds
    from oracle.vunet_oracle import make_synthetic_vgg19
    seed, K, total_steps = 53, 7, 8
    cfg = dict(ORG_CFG)
    net = rv.VunetOrg(n_channels_x=3, **cfg)
    sh = load_synth(net, seed)
    vsd = make_synthetic_vgg19(seed=79, width_div=8)
    pv = rp.PerceptualVGG(build_vgg_features(vsd), [1.0] * 6)
    pv.eval()
    lr0, betas, kl_init, kl_max, ll_weight = 8e-4, (0.5, 0.9), 1e-6, 1.0, 5.0    # config/vunet.yaml
    adjust_lr = partial(ru.linear_var, start_it=0, end_it=total_steps, start_val=lr0, end_val=0, clip_min=0, clip_max=lr0)
    adjust_kl = partial(ru.linear_var, start_it=total_steps // 2, end_it=3 * total_steps // 4, start_val=kl_init,
                        end_val=kl_max, clip_min=kl_init, clip_max=1.0)
    opt = torch.optim.Adam([{"params": getattr(net, n).parameters(), "name": n} for n in ["eu", "ed", "du", "dd"]],
                           lr=lr0, betas=betas)
    kl_weight, lr = float(adjust_kl(0)), float(adjust_lr(0))
    for pg_ in opt.param_groups:
        pg_["lr"] = lr
    rec, eps_shapes = [], None
    net.train()
    for it in range(1, K + 1):
        x = synth_image(f"otraj.x{it}", (2, 3, 32, 32), seed)
        c = synth_image(f"otraj.c{it}", (2, 3, 32, 32), seed)
        opt.zero_grad()
        with FixedNoise(f"otraj.{it}", seed) as fn:
            img, qs, ps, _ = net(x, c)
        eps_shapes = fn.shapes
        ld = rl.vgg_loss(pv, x, img)
        ll = ll_weight * torch.sum(torch.stack([ld[k] for k in ld], dim=0))
        kl = rl.compute_kl_loss(ps, qs)
        loss = ll + kl_weight * kl
        loss.backward()
        opt.step()
        rec.append({"it": it, "loss": float(loss), "ll": float(ll), "kl": float(kl), "kl_weight": kl_weight, "lr": lr})
        lr, kl_weight = float(adjust_lr(it)), float(adjust_kl(it))
        for pg_ in opt.param_groups:
            pg_["lr"] = lr
    csum = {k: [float(v.double().sum()), float(v.double().abs().sum())] for k, v in net.state_dict().items()}
    meta = {"seed": seed, "cfg": cfg, "shapes": sh, "vgg_seed": 79, "vgg_width_div": 8, "K": K, "lr0": lr0,
            "betas": list(betas), "kl_init": kl_init, "kl_max": kl_max, "ll_weight": ll_weight, "total_steps": total_steps,
            "eps_shapes": eps_shapes, "steps": rec, "param_checksums": csum}
    save("g5_org_trajectory", meta, {"final.dd.out_conv.conv.weight_v": net.state_dict()["dd.out_conv.conv.weight_v"].numpy()})


# ---------------------------------------------------------------- G5b trajectory with the regressor side loop on
def g5_regressor_trajectory():
    """experiments/shape_and_pose_net.py:360-466 with ``train_regressor: True`` driven by hand on the reference's
    modules: per outer step, reg_steps regressor Adam(lr 1e-3) steps on ``ed(eu(reg_imgs[:, i]))`` under no_grad
    (:407-425), then ``loss -= clamp(loss_regressor, max=1.2) * weight_regressor`` (:423-425)."""
    from oracle.vunet_oracle import make_synthetic_vgg19
    seed, K, R = 53, 3, 5
    cfg = dict(ALTER_CFG)
    net = rv.VunetAlter(n_channels_x=3, **cfg)
    sh = load_synth(net, seed)
    reg = rv.Regressor(n_out=34, n_latent_scales=2, nf_max=16, latent_widths=[8, 4], linear_width_factor=1)
    rsh = load_synth(reg, seed + 1)
    vsd = make_synthetic_vgg19(seed=77, width_div=8)
    pv = rp.PerceptualVGG(build_vgg_features(vsd), [1.0] * 6)
    pv.eval()
    lr0, betas, gamma_step, imax, n_init, total_steps, w_reg = 5e-4, (0.5, 0.9), 1e-5, 1000.0, 1, 100, 4.0
    opt = torch.optim.Adam([{"params": getattr(net, n).parameters(), "name": n} for n in ["eu", "ed", "du", "dd"]],
                           lr=lr0, betas=betas)
    opt_reg = torch.optim.Adam(reg.parameters(), lr=0.001)
    gamma, lr = 0.5, lr0
    rec = []
    net.train()
    for it in range(1, K + 1):
        x = synth_image(f"rtraj.x{it}", (2, 3, 32, 32), seed)
        c = synth_image(f"rtraj.c{it}", (2, 3, 32, 32), seed)
        reg_imgs = synth_image(f"rtraj.r{it}", (2, R, 3, 32, 32), seed)
        reg_targets = seeded_randn(f"rtraj.t{it}", (2, R, 17, 2), seed) * 0.25 + 0.5
        with FixedNoise(f"rtraj.{it}", seed):
            img, means, logstds, _ = net(x, c)
        ld = rl.vgg_loss(pv, x, img)
        ll = 1.0 * torch.sum(torch.stack([ld[k] for k in ld], dim=0))
        kl = rl.compute_kl_with_prior(means, logstds)
        loss = ll
        if it > n_init:
            loss = loss + torch.tensor(gamma) * kl
        reg_losses = []
        for i in range(R):
            with torch.no_grad():
                with FixedNoise(f"rtraj.{it}.reg{i}", seed):
                    _, rmeans, _, _ = net.ed(net.eu(reg_imgs[:, i]))
            preds = reg(rmeans)
            tgts = reg_targets[:, i].reshape(reg_targets[:, i].shape[0], -1)
            loss_regressor = torch.norm(preds - tgts, dim=1).mean()
            opt_reg.zero_grad()
            loss_regressor.backward(retain_graph=True)
            opt_reg.step()
            reg_losses.append(float(loss_regressor))
        # the reference subtracts the un-detached regressor loss; on torch >= 1.5 backward() then raises (the
        # regressor weights were stepped in place), on its own torch 1.3.1 it only produced regressor gradients that
        # the next zero_grad() discards.  No gradient reaches the VUnet either way (:413 runs under no_grad): detached,
        # the recorded values are the same.
        loss = loss - torch.clamp(loss_regressor.detach(), max=1.2) * w_reg
        opt.zero_grad()
        loss.backward()
        opt.step()
        gamma = max(gamma - gamma_step * (imax - float(kl)), 0)
        rec.append({"it": it, "loss": float(loss), "ll": float(ll), "kl": float(kl), "gamma_after": gamma, "lr": lr,
                    "reg_losses": reg_losses})
        lr = float(np.clip(float(0 - lr0) / (total_steps - 0) * (it - 0) + lr0, 0, lr0))
        for pg_ in opt.param_groups:
            pg_["lr"] = lr
    csum = {k: [float(v.double().sum()), float(v.double().abs().sum())] for k, v in net.state_dict().items()}
    rsum = {k: [float(v.double().sum()), float(v.double().abs().sum())] for k, v in reg.state_dict().items()}
    meta = {"seed": seed, "cfg": cfg, "shapes": sh, "reg_shapes": rsh, "reg_seed": seed + 1, "vgg_seed": 77,
            "vgg_width_div": 8, "K": K, "reg_steps": R, "lr0": lr0, "betas": list(betas), "gamma0": 0.5,
            "gamma_step": gamma_step, "imax": imax, "n_init_batches": n_init, "total_steps": total_steps,
            "weight_regressor": w_reg, "steps": rec, "param_checksums": csum, "reg_checksums": rsum}
    save("g5_regressor_trajectory", meta,
         {"final.dd.out_conv.conv.weight_v": net.state_dict()["dd.out_conv.conv.weight_v"].numpy(),
          "final.reg.linears.1.weight": reg.state_dict()["linears.1.weight"].numpy()})


# ---------------------------------------------------------------- G6 full-size sanity (statistics + crops, no full tensors)
FULL_CFGS = {
    # BASELINE config 2: Human3.6m, 256^2 (config/shape_and_pose_net.yaml:8-37)
    "h36m256": (dict(spatial_size=256, bottleneck_factor=2, box_factor=2, n_scales=0, n_latent_scales=2,
                     conv_layer_type="l1", nf_start=32, nf_max=128, subpixel_upsampling=True, dropout_prob=0.0),
                3, (2, 3, 256, 256), (2, 3, 256, 256)),
    # BASELINE config 1: Market, 128^2, 30-channel 64x64 appearance input (README.md:103-110, data/market.py:45)
    "market128": (dict(spatial_size=128, bottleneck_factor=1, box_factor=1, n_scales=0, n_latent_scales=2,
                       conv_layer_type="l1", nf_start=32, nf_max=128, subpixel_upsampling=True, dropout_prob=0.0),
                  30, (2, 30, 64, 64), (2, 3, 128, 128)),
}
FULL_GRAD_KEYS = ["eu.nin.conv.weight_v", "eu.blocks.0.conv.conv.weight_v", "eu.downs.0.down.conv.weight_v",
                  "ed.make_logstds.1.conv.weight_g", "ed.blocks.1.nin.conv.weight_v", "du.blocks.3.conv.gamma",
                  "dd.ups.0.up.conv.weight_v", "dd.blocks.11.conv.conv.weight_v", "dd.auto_blocks.1.conv.beta",
                  "dd.out_conv.conv.weight_v", "dd.out_conv.conv.bias"]


def g6_full_size():
    seed = 61
    for tag, (cfg, ncx, xshape, cshape) in FULL_CFGS.items():
        net = rv.VunetAlter(n_channels_x=ncx, **cfg)
        sh = load_synth(net, seed)
        net.train()
        x, c = synth_image(tag + ".x", xshape, seed), synth_image(tag + ".c", cshape, seed)
        with FixedNoise(tag, seed) as fn:
            img, means, logstds, _ = net(x, c)
        wgt = seeded_randn(tag + ".w", tuple(img.shape), seed)
        (img * wgt).sum().backward()
        params = dict(net.named_parameters())
        keys = [k for k in FULL_GRAD_KEYS if k in params]
        arrays = {"img_crop": img.detach()[:, :, 40:56, 72:88].numpy(),
                  "img_sum": img.detach().double().sum(dim=(2, 3)).numpy(),
                  "img_abssum": img.detach().double().abs().sum(dim=(2, 3)).numpy(),
                  "img_max": img.detach().amax(dim=(2, 3)).numpy(), "img_min": img.detach().amin(dim=(2, 3)).numpy()}
        for i, (m, l) in enumerate(zip(means, logstds)):
            arrays[f"mean{i}_sum"] = m.detach().double().sum(dim=(2, 3)).numpy()
            arrays[f"logstd{i}_sum"] = l.detach().double().sum(dim=(2, 3)).numpy()
        for k in keys:
            g = params[k].grad
            arrays["gp." + k] = g.reshape(-1)[:4096].numpy()
        gsum = {k: (None if p.grad is None else [float(p.grad.double().sum()), float(p.grad.double().abs().sum()),
                                                 float(p.grad.abs().max())]) for k, p in params.items()}
        meta = {"seed": seed, "cfg": cfg, "n_channels_x": ncx, "x": list(xshape), "c": list(cshape),
                "eps_shapes": fn.shapes, "n_params": sum(p.numel() for p in net.parameters()), "n_keys": len(sh),
                "crop": [40, 56, 72, 88], "grad_sums": gsum}
        save("g6_" + tag, meta, arrays)


# ---------------------------------------------------------------- G1b the bilinear Upsample branch (lib/modules.py:172-182)
def g1b_upsample_bilinear():
    seed = 13
    arrays, meta = {}, {"seed": seed}
    up = rm.Upsample(8, 6, subpixel=False)
    meta["shapes"] = load_synth(up, seed)
    up.train()
    x = synth_image("upb.x", (2, 8, 7, 10), seed).requires_grad_(True)
    y = up(x)
    (y * seeded_randn("upb.w", tuple(y.shape), seed)).sum().backward()
    arrays["y"], arrays["gx"] = y.detach().numpy(), x.grad.numpy()
    for k, p_ in up.named_parameters():
        arrays["gp." + k] = p_.grad.numpy()
    # whole model with subpixel_upsampling False: sub-pixel on the latent levels, bilinear past them (models/vunets.py:325-329)
    cfg = dict(ALTER_CFG)
    cfg["subpixel_upsampling"] = False
    net = rv.VunetAlter(n_channels_x=3, **cfg)
    meta["model_shapes"] = load_synth(net, seed)
    meta["cfg"] = cfg
    net.train()
    xi, c = synth_image("upb.mx", (2, 3, 32, 32), seed), synth_image("upb.mc", (2, 3, 32, 32), seed)
    with FixedNoise("upb", seed) as fn:
        img, means, logstds, _ = net(xi, c)
    (img * seeded_randn("upb.mw", tuple(img.shape), seed)).sum().backward()
    arrays["img"] = img.detach().numpy()
    meta["eps_shapes"] = fn.shapes
    meta["grad_sums"] = {k: (None if p_.grad is None else [float(p_.grad.double().sum()), float(p_.grad.double().abs().sum())])
                         for k, p_ in net.named_parameters()}
    save("g1b_upsample_bilinear", meta, arrays)


# ---------------------------------------------------------------- G1c L2NormConv2d data-dependent init (lib/modules.py:95-99)
def g1c_l2norm_init():
    """``init_fn() -> True`` in training mode: the first forward sets gamma = 1 / sqrt(var + 1e-10), beta = -mean * gamma from
    the batch statistics of the normalised convolution over (N, H, W) and returns gamma * x + beta with them; a second
    forward (init over) uses the stored values.  As wired by models/vunets.py:449: bias=False."""
    seed = 17
    arrays, meta = {}, {"seed": seed, "cases": {}}
    flag = {"on": True}
    for case, (cin, cout, k, s_, p_, shape) in {"k3": (6, 8, 3, 1, 1, (3, 6, 8, 8)), "k3s2": (8, 16, 3, 2, 1, (2, 8, 12, 12)),
                                                "k1": (5, 4, 1, 1, 0, (2, 5, 6, 6))}.items():
        mod = rm.L2NormConv2d(cin, cout, k, s_, p_, bias=False, init=lambda: flag["on"])
        sh = load_synth(mod, seed)
        mod.train()
        flag["on"] = True
        x = synth_image(f"l2i.{case}.x", shape, seed)
        y1 = mod(x)
        arrays[f"{case}.y_init"] = y1.detach().numpy()
        arrays[f"{case}.gamma"] = mod.gamma.detach().numpy()
        arrays[f"{case}.beta"] = mod.beta.detach().numpy()
        flag["on"] = False
        x2 = synth_image(f"l2i.{case}.x2", shape, seed)
        arrays[f"{case}.y_after"] = mod(x2).detach().numpy()
        meta["cases"][case] = {"shapes": sh, "args": [cin, cout, k, s_, p_], "input": list(shape)}
    save("g1c_l2norm_init", meta, arrays)


# ---------------------------------------------------------------- G8 a pretrained-model directory as the reference writes it
def g8_pretrained_dir():
    """main.py:38-47 / experiments/experiment.py:39-95, experiments/shape_and_pose_net.py:471-482: a directory holding
    ``config.yaml`` (the reference's own YAML with the sizes of a small model) and ``reg_ckpt_*_<it>.pth`` files
    ``{"model": VunetAlter.state_dict(), "optimizer": torch.optim.Adam.state_dict()}`` -- written HERE by the reference's
    classes and torch.optim.Adam after two training steps.  Recorded with it: what the restored reference model computes (``transfer``) on a seeded pair."""
    import yaml
    from oracle.vunet_oracle import make_synthetic_vgg19
    seed = 81
    with open(os.path.join(REF, "config", "shape_and_pose_net.yaml")) as f:
        cdict = yaml.load(f, Loader=yaml.FullLoader)
    cdict["data"].update(spatial_size=32)
    cdict["architecture"].update(nf_start=4, nf_max=8)
    cdict["training"].update(dropout_prob=0.0, train_regressor=False)
    cdict["general"].update(base_dir="/tmp/vunet_pretrained_fixture")
    out_dir = os.path.join(HERE, "g8_pretrained")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "config.yaml"), "w") as f:
        yaml.dump(cdict, f, default_flow_style=False)
    kw = dict(cdict["architecture"])
    kw.update(cdict["data"])
    kw["dropout_prob"] = cdict["training"]["dropout_prob"]
    torch.manual_seed(seed)
    net = rv.VunetAlter(n_channels_x=3, **kw)          # the reference's default initialisation
    vsd = make_synthetic_vgg19(seed=77, width_div=8)
    pv = rp.PerceptualVGG(build_vgg_features(vsd), [1.0] * 6)
    pv.eval()
    opt = torch.optim.Adam([{"params": getattr(net, n).parameters(), "name": n} for n in ["eu", "ed", "du", "dd"]],
                           lr=cdict["training"]["lr"], betas=tuple(cdict["training"]["adam_betas"]))
    net.train()
    for it in (1, 2):
        x, c = synth_image(f"pre.x{it}", (2, 3, 32, 32), seed), synth_image(f"pre.c{it}", (2, 3, 32, 32), seed)
        with FixedNoise(f"pre.{it}", seed):
            img, means, logstds, _ = net(x, c)
        ld = rl.vgg_loss(pv, x, img)
        loss = torch.sum(torch.stack([ld[k] for k in ld], dim=0)) + 0.01 * rl.compute_kl_with_prior(means, logstds)
        opt.zero_grad()
        loss.backward()
        opt.step()
        for pg_ in opt.param_groups:      # :507-512: gamma rides in the param groups
            pg_["gamma"] = 0.25 * it
        if it == 2:   # (one file is committed; the test puts an older decoy beside it for the selection rule)
            torch.save({"model": net.state_dict(), "optimizer": opt.state_dict()},
                       os.path.join(out_dir, f"reg_ckpt_model_{it}.pth"))
    net.eval()
    x, c = synth_image("pre.tx", (2, 3, 32, 32), seed), synth_image("pre.tc", (2, 3, 32, 32), seed)
    with torch.no_grad():
        with FixedNoise("pre.t", seed):
            img = net.transfer(x, c)
    meta = {"seed": seed, "newest": "reg_ckpt_model_2.pth", "iteration": 2, "gamma": 0.5,
            "n_tensors": len(net.state_dict()),
            "checksums": {k: [float(v.double().sum()), float(v.double().abs().sum())] for k, v in net.state_dict().items()}}
    save("g8_pretrained_outputs", meta, {"transfer": img.numpy()})


# ---------------------------------------------------------------- G7 evaluation statistics (lib/metrics.py:277-415)
def g7_metrics():
    """FID statistics (_calculate_fid on mean / np.cov of features) and the Inception score (inception_score driven with
    a small seeded linear "classifier" in place of torchvision's inception_v3, which is absent): the reference's own
    functions on synthetic features / images."""
    for name in ["skimage", "skimage.metrics", "h5py", "imagesize", "umap"]:
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)
                sys.modules[name].__getattr__ = lambda k: type(k, (), {})
    from lib import metrics as rmet
    seed = 71
    arrays, meta = {}, {"seed": seed}
    # FID: two feature clouds with different means / covariances, D = 24
    a = seeded_randn("fid.a", (400, 24), seed).double().numpy() @ seeded_randn("fid.ma", (24, 24), seed).double().numpy()
    b = seeded_randn("fid.b", (300, 24), seed).double().numpy() @ seeded_randn("fid.mb", (24, 24), seed).double().numpy() + 0.3
    mu1, c1 = np.mean(a, axis=0), np.cov(a, rowvar=False)
    mu2, c2 = np.mean(b, axis=0), np.cov(b, rowvar=False)
    meta["fid"] = float(rmet._calculate_fid(mu1, c1, mu2, c2))
    meta["fid_same"] = float(rmet._calculate_fid(mu1, c1, mu1, c1))
    # rank-deficient covariance (fewer samples than dimensions): the eps branch / complex sqrtm handling
    d = seeded_randn("fid.d", (10, 24), seed).double().numpy()
    mu3, c3 = np.mean(d, axis=0), np.cov(d, rowvar=False)
    meta["fid_singular"] = float(rmet._calculate_fid(mu1, c1, mu3, c3))
    # Inception score: reference function, fake classifier = seeded linear map of the 8x8-average-pooled image
    w = seeded_randn("is.w", (3 * 8 * 8, 1000), seed) * 3.0   # 1000 classes: the reference hard-codes preds[N, 1000]

    class Fake(torch.nn.Module):
        def forward(self, x):
            return torch.nn.functional.adaptive_avg_pool2d(x, 8).flatten(1) @ w
    rmet.inception_v3 = lambda **kw: Fake()
    rmet.tqdm = lambda it, **kw: it
    imgs = synth_image("is.imgs", (96, 3, 32, 32), seed)
    ds = torch.utils.data.TensorDataset(imgs)
    for splits in (1, 4):
        m, sd_ = rmet.inception_score(ds, torch.device("cpu"), batch_size=16, resize=False, splits=splits)
        meta[f"is_{splits}"] = [float(m), float(sd_)]
    save("g7_metrics", meta, arrays)


def behavior_state(mod, seed):
    """Seeded values for every floating tensor of a behaviour / flow module (synth.synth_behavior_state); index buffers
    (``Shuffle``) keep the permutation the reference drew."""
    sd = mod.state_dict()
    stored = {k: v for k, v in sd.items() if not v.dtype.is_floating_point and k.rsplit(".", 1)[-1] != "initialized"}
    out = synth_behavior_state({k: list(v.shape) for k, v in sd.items()}, seed, stored)
    mod.load_state_dict(out)
    return out


def g9_behavior():
    """Behaviour front half of config 5 (experiments/behavior_net.py:1086-1100, :1173-1184): the reference's own
    ``UnsupervisedTransformer2`` (both directions) and ``ResidualBehaviorNet`` (``infer_b``, ``generate_seq``,
    ``forward``) at small sizes, one flow with an odd channel count.  The fixture stores the permutations the
    reference drew, the recipe of every other tensor and the expected outputs."""
    from models import pose_behavior_rnn as rb
    from models.flow import simple_flow as rf
    seed = 91
    arrays, meta = {}, {"seed": seed, "cases": {}}
    for tag, (chan, mid, depth, n_flows, bsz) in {"even": (64, 128, 2, 3, 5), "odd": (33, 48, 1, 2, 4)}.items():
        torch.manual_seed(seed)
        flow = rf.UnsupervisedTransformer2(flow_in_channels=chan, flow_mid_channels=mid, flow_hidden_depth=depth,
                                           n_flows=n_flows)
        sd = behavior_state(flow, seed)
        flow.eval()
        x = seeded_randn(f"flow.{tag}.x", (bsz, chan), seed)
        z = seeded_randn(f"flow.{tag}.z", (bsz, chan), seed)
        with torch.no_grad():
            out, logdet = flow(x)
            rev = flow.reverse(z)
        meta["cases"][f"flow_{tag}"] = {"kw": dict(flow_in_channels=chan, flow_mid_channels=mid, flow_hidden_depth=depth,
                                                   n_flows=n_flows), "batch": bsz,
                                        "shapes": {k: list(v.shape) for k, v in sd.items()}}
        for k, v in sd.items():
            if k.endswith("_shuffle_idx"):
                arrays[f"flow_{tag}.sd.{k}"] = v.numpy()
        arrays[f"flow_{tag}.forward"] = out.reshape(bsz, chan).numpy()
        arrays[f"flow_{tag}.logdet"] = logdet.numpy()
        arrays[f"flow_{tag}.reverse"] = rev.reshape(bsz, chan).numpy()
    # the behaviour net: information bottleneck on, LSTM decoder, with and without the decoder's input layer
    n_kps, hid, bsz, t_in, length = 51, 64, 5, 7, 6
    for tag, nin in {"plain": False, "nin": True}.items():
        torch.manual_seed(seed)
        net = rb.ResidualBehaviorNet(n_kps=n_kps, information_bottleneck=True, decoder_arch="lstm",
                                     linear_in_decoder=nin, dim_hidden_b=hid)
        sd = behavior_state(net, seed)
        net.eval()
        net.b_enc.init_hidden = lambda bs, device: rb.BEncoder.init_hidden(net.b_enc, bs, "cpu")   # get_device() is -1 on CPU
        x1 = 0.5 * seeded_randn(f"net.{tag}.x1", (bsz, t_in, n_kps), seed)
        x2 = 0.5 * seeded_randn(f"net.{tag}.x2", (bsz, t_in, n_kps), seed)
        b_given = seeded_randn(f"net.{tag}.b", (bsz, hid), seed)
        with torch.no_grad():
            gen_xs, gen_cs, _, _ = net.generate_seq(b_given, x2, len=length, start_frame=t_in - 1)
            with FixedNoise(f"net.{tag}", seed) as fn:
                xs, cs, _, b, mu, logstd, pre = net(x1, x2, length, start_frame=2)
            with FixedNoise(f"net.{tag}.prior", seed):
                xs_p, _, _, b_p, *_ = net(x1, x2, length, start_frame=0, sample=True)
        meta["cases"][f"net_{tag}"] = {"kw": dict(n_kps=n_kps, information_bottleneck=True, decoder_arch="lstm",
                                                  linear_in_decoder=nin, dim_hidden_b=hid),
                                       "batch": bsz, "t_in": t_in, "len": length, "noise_shapes": fn.shapes,
                                       "shapes": {k: list(v.shape) for k, v in sd.items()}}
        for name, v in dict(gen_xs=gen_xs, gen_cs=gen_cs, xs=xs, cs=cs, b=b, mu=mu, logstd=logstd, pre=pre,
                            xs_prior=xs_p, b_prior=b_p).items():
            arrays[f"net_{tag}.{name}"] = v.numpy()
    save("g9_behavior", meta, arrays)


def g9b_projection():
    """The numpy chain between the decoder's pose vectors and the rasteriser's pixel keypoints, from the reference's own
    functions: ``unNormalizeData`` (data/data_conversions_3d.py:178-211), ``apply_affine_transform`` (:588-605),
    ``camera_projection`` (:892-912) and the joint rescale of the render loop (:1132-1133, :1149-1150).  Two statistics dtypes
    (numpy's promotion differs: float32 statistics keep the un-normalisation in float32), ignored dimensions, a rectangular
    source image.  ``data.data_conversions_3d`` imports the dataset stack at module level: the absent third-party packages
    behind it are stubbed; the three functions are pure numpy."""
    for name in ["h5py", "imagesize", "skimage", "skimage.metrics", "natsort", "kornia.geometry", "umap"]:
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)
                sys.modules[name].__getattr__ = lambda k: type(k, (), {"__init__": lambda self, *a, **kw: None})
    from data import data_conversions_3d as dc
    arrays, meta = {}, {"cases": {}}
    for tag, (dims, n_ignore, stats_dtype) in {"h36m_f32": (96, 45, "float32"), "h36m_f64": (96, 45, "float64"),
                                               "plain": (51, 3, "float64")}.items():
        rng = np.random.RandomState(17 + dims)
        ignore = sorted(rng.choice(dims, size=n_ignore, replace=False).tolist())
        mean = (rng.randn(dims) * 300.0).astype(stats_dtype)
        std = (50.0 + 200.0 * rng.rand(dims)).astype(stats_dtype)
        ang = 0.4
        rot = np.array([[np.cos(ang), 0.0, np.sin(ang)], [0.0, 1.0, 0.0], [-np.sin(ang), 0.0, np.cos(ang)]])
        ext = np.concatenate([rot, np.array([[50.0], [-120.0], [5200.0]])], axis=1)
        intr = (1145.0, 512.5, 1143.8, 515.4)
        image_size, spatial = (1000, 1002), 256
        x = rng.randn(23, dims - n_ignore).astype(np.float32)
        poses = dc.revert_output_format(x, mean, std, ignore).reshape(23, -1, 3)                 # experiments/behavior_net.py:1181-1183
        size_arr = np.full((1, 2), spatial, dtype=float)
        out = []
        for p_ in poses:
            pose_c = dc.apply_affine_transform(p_, ext)
            pose_i = dc.camera_projection(pose_c, intr)
            out.append(pose_i * (size_arr / np.expand_dims(np.asarray(image_size, dtype=float), axis=0)))
        meta["cases"][tag] = {"dims": dims, "ignore": ignore, "stats_dtype": stats_dtype, "intrinsics": list(intr),
                              "image_size": list(image_size), "spatial_size": spatial}
        arrays[f"{tag}.x"], arrays[f"{tag}.mean"], arrays[f"{tag}.std"], arrays[f"{tag}.ext"] = x, mean, std, ext
        arrays[f"{tag}.kps"] = np.stack(out)
    save("g9b_projection", meta, arrays)


def g10_flow_training():
    """BASELINE config 4, flow stage (experiments/behavior_net.py:384-395, :703-714): the reference's own
    ``UnsupervisedTransformer2`` + ``FlowLoss`` (lib/losses.py:294-317) + ``torch.optim.Adam(betas=(0.5, 0.9), weight_decay)``
    driven for K = 3 steps exactly as ``train_fn`` drives them -- ``gauss, logdet = latent_flow(bs)``; ``flow_loss``;
    ``zero_grad``; ``backward``; ``step`` -- at small sizes: an even channel count with ActNorm's data-dependent
    initialisation happening inside step 1 (a fresh flow, as the reference trains it), and an odd one, pre-initialised, with
    weight decay.  Stored: the permutations, every step's log, the parameters and two moments after the last step."""
    from models.flow import simple_flow as rf
    seed = 101
    arrays, meta = {}, {"seed": seed, "steps": 3, "cases": {}}
    cases = {"even": dict(chan=32, mid=64, depth=2, n_flows=3, bsz=16, lr=3e-4, wd=0.0, fresh=True),
             "odd": dict(chan=33, mid=48, depth=1, n_flows=2, bsz=4, lr=3e-4, wd=1e-2, fresh=False)}
    for tag, c in cases.items():
        torch.manual_seed(seed)
        flow = rf.UnsupervisedTransformer2(flow_in_channels=c["chan"], flow_mid_channels=c["mid"], flow_hidden_depth=c["depth"],
                                           n_flows=c["n_flows"])
        sd = behavior_state(flow, seed)
        if c["fresh"]:    # ActNorm as a fresh flow has it: loc 0, scale 1, not initialised (lib/modules.py:264-268)
            for k in list(sd):
                leaf = k.rsplit(".", 1)[-1]
                if leaf == "initialized":
                    sd[k] = torch.tensor(0, dtype=torch.uint8)
                elif leaf == "loc":
                    sd[k] = torch.zeros_like(sd[k])
                elif leaf == "scale" and ".norm_layer." in k:
                    sd[k] = torch.ones_like(sd[k])
            flow.load_state_dict(sd)
        flow.train()
        opt = torch.optim.Adam(params=[{"params": flow.parameters(), "name": "latent_flow"}], lr=c["lr"], betas=(0.5, 0.9),
                               weight_decay=c["wd"])
        loss_fn = rl.FlowLoss()
        logs = []
        for it in range(meta["steps"]):
            bs = 0.8 * seeded_randn(f"flowtrain.{tag}.b{it}", (c["bsz"], c["chan"]), seed) + 0.3
            with FixedNoise(f"flowtrain.{tag}.s{it}", seed):
                gauss, logdet = flow(bs.detach())
                f_loss, log = loss_fn(gauss, logdet)
            opt.zero_grad()
            f_loss.backward()
            opt.step()
            logs.append({k: float(v) for k, v in log.items()})
        names = [n for n, _ in flow.named_parameters()]
        fin = flow.state_dict()
        meta["cases"][tag] = {"kw": dict(flow_in_channels=c["chan"], flow_mid_channels=c["mid"], flow_hidden_depth=c["depth"],
                                         n_flows=c["n_flows"]), "batch": c["bsz"], "lr": c["lr"], "weight_decay": c["wd"],
                              "fresh": c["fresh"], "logs": logs, "shapes": {k: list(v.shape) for k, v in sd.items()},
                              "checksums": {k: [float(v.double().sum()), float(v.double().abs().sum())] for k, v in fin.items()
                                            if v.dtype.is_floating_point}}
        for k, v in sd.items():
            if k.endswith("_shuffle_idx"):
                arrays[f"{tag}.sd.{k}"] = v.numpy()
        keep = [n for n in names if ".norm_layer." in n or n.endswith(".bias")
                or n.startswith("flow.sub_layers.0.coupling.s.0.") or n.startswith(f"flow.sub_layers.{c['n_flows'] - 1}.coupling.t.1.")]
        if tag == "odd":
            keep = names
        for n in keep:
            arrays[f"{tag}.final.{n}"] = fin[n].numpy()
        st = opt.state_dict()["state"]
        for n in (f"flow.sub_layers.0.coupling.s.0.main.0.weight", f"flow.sub_layers.{c['n_flows'] - 1}.norm_layer.scale"):
            idx = names.index(n)
            arrays[f"{tag}.exp_avg.{n}"] = st[idx]["exp_avg"].numpy()
            arrays[f"{tag}.exp_avg_sq.{n}"] = st[idx]["exp_avg_sq"].numpy()
        meta["cases"][tag]["adam_step"] = int(st[0]["step"])
    save("g10_flow_training", meta, arrays)


def g11_cvae_training():
    """BASELINE config 4, first stage (experiments/behavior_net.py:591-660 with ``get_loss`` :134-149, ``kl_loss``
    lib/losses.py:283-291, the gamma controller :111-116 and ``Adam(to_optim, lr=lr_init)`` :329-336): the reference's own
    ``ResidualBehaviorNet`` driven for K = 3 steps as ``train_fn`` drives it -- ``net(seq_b, seq_b, seq_len)``, recon =
    mean(MSELoss(reduction none)), ``loss = recon_loss_weight * recon + gamma * kl``, ``zero_grad`` / ``backward`` / ``step``,
    then the gamma update -- with recorded reparametrisation noise.  ``use_regressor`` is off: with it on the reference's
    step cannot run on torch >= 1.5 (the regressor's weights are stepped in place between the loss's forward and its
    backward, :636-653); the fixture records the error the reference raises here."""
    from models import pose_behavior_rnn as rb
    seed = 111
    n_kps, hid, bsz, t_len = 51, 64, 5, 6
    w_recon, gamma, gamma_step, imax, lr = 2.5, 0.05, 1e-3, 5.0, 1e-3
    torch.manual_seed(seed)
    net = rb.ResidualBehaviorNet(n_kps=n_kps, information_bottleneck=True, decoder_arch="lstm", linear_in_decoder=False,
                                 dim_hidden_b=hid)
    sd = behavior_state(net, seed)
    net.train()
    net.b_enc.init_hidden = lambda bs, device: rb.BEncoder.init_hidden(net.b_enc, bs, "cpu")   # get_device() is -1 on CPU
    opt = torch.optim.Adam([{"params": net.b_enc.parameters(), "name": "z_enc"}, {"params": net.decoder.parameters(), "name": "dec"}],
                           lr=lr)
    rec_loss = torch.nn.MSELoss(reduction="none")
    meta = {"seed": seed, "steps": 3, "kw": dict(n_kps=n_kps, information_bottleneck=True, decoder_arch="lstm", linear_in_decoder=False,
                                                 dim_hidden_b=hid),
            "batch": bsz, "seq_len": t_len, "recon_loss_weight": w_recon, "gamma_init": gamma, "gamma_step": gamma_step, "imax": imax,
            "lr": lr, "shapes": {k: list(v.shape) for k, v in sd.items()}, "logs": []}
    arrays = {}
    for it in range(meta["steps"]):
        kps = 0.5 * seeded_randn(f"cvae.kps{it}", (bsz, t_len + 1, n_kps), seed)
        seq_b, target = kps[:, :-1], kps[:, 1:]                      # prepare_input (lib/utils.py:914-917)
        with FixedNoise(f"cvae.s{it}", seed) as fn:
            xs, cs, _, bs, mu, logstd, pre = net(seq_b, seq_b, t_len)
        r = rec_loss(xs, target)
        recon, per_seq = torch.mean(r), torch.mean(r, dim=[0, 2])
        kl = rl.kl_loss(mu, logstd)
        loss = w_recon * recon + gamma * kl
        opt.zero_grad()
        loss.backward()
        opt.step()
        used = gamma
        gamma = max(gamma - gamma_step * (imax - float(kl)), 0)
        meta["logs"].append({"loss": float(loss), "loss_recon": float(recon), "kl_loss": float(kl), "gamma_used": used, "gamma": gamma,
                             "mu_s": float(torch.mean(mu)), "logstd_s": float(torch.mean(logstd))})
        arrays[f"per_seq{it}"] = per_seq.detach().numpy()
        if it == 0:
            arrays["xs0"], arrays["bs0"] = xs.detach().numpy(), bs.detach().numpy()
            # the gradients of the first step, before the update (zero_grad ran before backward)
    meta["noise_shapes"] = fn.shapes
    fin = net.state_dict()
    meta["checksums"] = {k: [float(v.double().sum()), float(v.double().abs().sum())] for k, v in fin.items()}
    for k, v in fin.items():
        arrays[f"final.{k}"] = (v[:24] if v.dim() == 2 and v.shape[0] > 64 else v).numpy()
    names = [n for n, _ in net.named_parameters()]
    st = opt.state_dict()["state"]
    for n in ("decoder.rnn.weight_hh", "b_enc.mu_fn.conv.weight_v", "decoder.n_out.bias"):
        e = st[names.index(n)]
        arrays[f"exp_avg.{n}"] = (e["exp_avg"][:24] if e["exp_avg"].dim() == 2 and e["exp_avg"].shape[0] > 64 else e["exp_avg"]).numpy()
    meta["adam_step"] = int(st[0]["step"])
    # the regressor path of the reference's step, as it stands, on this torch
    try:
        reg = rb.Regressor_fly(hid, n_kps)
        ropt = torch.optim.Adam(reg.parameters(), lr=1e-4)
        kps = 0.5 * seeded_randn("cvae.kpsr", (bsz, 51, n_kps), seed)
        seq_b = kps[:, :-1]
        with FixedNoise("cvae.sr", seed):
            xs, cs, _, bs, mu, logstd, pre = net(seq_b, seq_b, 50)
        loss = torch.mean(rec_loss(xs, kps[:, 1:]))
        for _ in range(5):
            idx = torch.randint(0, 50, (1,))
            oh = torch.nn.functional.one_hot(idx.repeat(mu.size(0)), num_classes=50)
            lreg = torch.mean((reg(mu, oh.float()) - seq_b[:, idx].squeeze()) ** 2)
            ropt.zero_grad()
            lreg.backward(retain_graph=True)
            ropt.step()
        loss -= torch.clamp(lreg, max=0.45) * 0.01
        loss.backward()
        meta["regressor_path"] = "ran"
    except RuntimeError as e:
        meta["regressor_path"] = "RuntimeError: " + str(e).split("\n")[0][:160]
    save("g11_cvae_training", meta, arrays)


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(4)
    only = sys.argv[1:]
    if only:   # e.g. `make_golden.py g6_full_size` regenerates one group
        for name in only:
            globals()[name]()
        sys.exit(0)
    g1_primitives()
    g2_models()
    g3_losses()
    g4_discriminators()
    g5_trajectory()
    g5_regressor_trajectory()
    g5_org_trajectory()
    g6_full_size()
    g7_metrics()
    g1b_upsample_bilinear()
    g1c_l2norm_init()
    g8_pretrained_dir()
    g9_behavior()
    g9b_projection()
    g10_flow_training()
    g11_cvae_training()
