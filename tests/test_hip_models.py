"""-m gpu: whole-model parity of the HIP product path against golden vectors produced by the reference
(tests/golden/make_golden.py) and against the CPU oracle at the real layer widths.

Tolerances (fp32): outputs atol/rtol 1e-4 (the fp32-vs-fp64 noise floor of the reference
itself is 5e-7 at 256^2), gradients rtol 1e-3 (long reductions), PSNR vs oracle >= 100 dB.
"""
import pytest
import torch

from conftest import load_golden
from hip_parity_utils import assert_close, psnr
from synth import seeded_randn, synth_image, synth_state_dict

pytestmark = pytest.mark.gpu


def _model_loss(outs, tag, seed):
    flat = []
    for o in outs:
        flat.extend(o) if isinstance(o, (list, tuple)) else flat.append(o)
    return sum((o * seeded_randn(f"{tag}.lw{i}", tuple(o.shape), seed).to(o.device)).sum() for i, o in enumerate(flat))


@pytest.mark.parametrize("tag", ["alter", "alter_box"])
def test_vunet_alter_vs_golden(tag):
    from behavior_driven_video_synthesis_amd.models.vunets import VunetAlter
    meta, arr = load_golden("g2_" + tag)
    seed, cfg, ncx = meta["seed"], meta["cfg"], meta["n_channels_x"]
    net = VunetAlter(n_channels_x=ncx, **cfg)
    assert {k: list(v.shape) for k, v in net.state_dict().items()} == meta["shapes"]
    net.load_state_dict(synth_state_dict(meta["shapes"], seed), strict=True)
    net = net.cuda().train()
    x = synth_image(tag + ".x", tuple(meta["x"]), seed).cuda().requires_grad_(True)
    c = synth_image(tag + ".c", tuple(meta["c"]), seed).cuda().requires_grad_(True)
    eps = [seeded_randn(f"{tag}.eps{i}", tuple(s), seed).cuda() for i, s in enumerate(meta["eps_shapes"])]
    img, means, logstds, acts = net(x, c, eps)
    assert_close(img, arr["img"], name="img")
    for i in range(len(means)):
        assert_close(means[i], arr[f"mean{i}"], name=f"mean{i}")
        assert_close(logstds[i], arr[f"logstd{i}"], name=f"logstd{i}")
    _model_loss([img, means, logstds], tag, seed).backward()
    assert_close(x.grad, arr["gx"], rtol=1e-3, atol=1e-5, name="gx")
    assert_close(c.grad, arr["gc"], rtol=1e-3, atol=1e-5, name="gc")
    params = dict(net.named_parameters())
    for k, v in arr.items():
        if k.startswith("gp."):
            assert_close(params[k[3:]].grad, v, rtol=1e-3, atol=2e-4, name=k)
    for k, s in meta["grad_sums"].items():
        g = params[k].grad
        if s is None:
            assert g is None or float(g.abs().sum()) == 0.0, k
        else:
            got = float(g.double().abs().sum())
            assert abs(got - s[1]) <= 1e-3 * s[1] + 1e-4, (k, got, s[1])
    net.eval()
    with torch.no_grad():
        eps_t = [seeded_randn(f"{tag}.tr.eps{i}", tuple(s), seed).cuda() for i, s in enumerate(meta["eps_shapes"])]
        assert_close(net.transfer(x, c, eps_t), arr["transfer"], name="transfer")
        pe = [seeded_randn(f"{tag}.tf.eps{i}", tuple(s), seed).cuda() for i, s in enumerate(meta["tf_eps_shapes"])]
        assert_close(net.test_forward(c, pe), arr["test_forward"], name="test_forward")


def test_vunet_org_vs_golden():
    from behavior_driven_video_synthesis_amd.models.vunets import VunetOrg
    from behavior_driven_video_synthesis_amd.lib.losses import compute_kl_loss
    tag = "org"
    meta, arr = load_golden("g2_org")
    seed, cfg = meta["seed"], meta["cfg"]
    net = VunetOrg(n_channels_x=3, **cfg)
    assert {k: list(v.shape) for k, v in net.state_dict().items()} == meta["shapes"]
    net.load_state_dict(synth_state_dict(meta["shapes"], seed))
    net = net.cuda().train()
    x = synth_image(tag + ".x", tuple(meta["x"]), seed).cuda().requires_grad_(True)
    c = synth_image(tag + ".c", tuple(meta["c"]), seed).cuda().requires_grad_(True)
    shapes, n_lat = meta["eps_shapes"], cfg["n_latent_scales"]
    eps = [seeded_randn(f"{tag}.eps{i}", tuple(shapes[i]), seed).cuda() for i in range(n_lat)]
    prior = [[seeded_randn(f"{tag}.eps{n_lat + 4 * i + l}", tuple(shapes[n_lat + 4 * i + l]), seed).cuda()
              for l in range(4)] for i in range(n_lat)]
    img, qs, ps, _ = net(x, c, eps, prior)
    assert_close(img, arr["img"], name="img")
    for i in range(n_lat):
        assert_close(qs[i], arr[f"q{i}"], name=f"q{i}")
        assert_close(ps[i], arr[f"p{i}"], name=f"p{i}")
    assert_close(compute_kl_loss(ps, qs), arr["kl"], rtol=1e-4, name="kl")
    _model_loss([img, qs, ps], tag, seed).backward()
    assert_close(x.grad, arr["gx"], rtol=1e-3, atol=1e-5, name="gx")
    assert_close(c.grad, arr["gc"], rtol=1e-3, atol=1e-5, name="gc")
    params = dict(net.named_parameters())
    for k, s in meta["grad_sums"].items():
        if s is not None:
            got = float(params[k].grad.double().abs().sum())
            assert abs(got - s[1]) <= 1e-3 * s[1] + 1e-4, (k, got, s[1])


def test_regressor_vs_golden():
    from behavior_driven_video_synthesis_amd.models.vunets import Regressor
    meta, arr = load_golden("g2_regressor")
    reg = Regressor(n_out=34, n_latent_scales=2, nf_max=16, latent_widths=[8, 4], linear_width_factor=1)
    assert {k: list(v.shape) for k, v in reg.state_dict().items()} == meta["shapes"]
    reg.load_state_dict(synth_state_dict(meta["shapes"], meta["seed"]))
    reg = reg.cuda()
    e0 = seeded_randn("reg.e0", (2, 16, 4, 4), meta["seed"]).cuda()
    e1 = seeded_randn("reg.e1", (2, 16, 8, 8), meta["seed"]).cuda()
    assert_close(reg([e0, e1]), arr["out"], name="regressor")


def test_losses_and_vgg_vs_golden():
    from behavior_driven_video_synthesis_amd.lib.losses import compute_kl_loss, compute_kl_with_prior, vgg_loss
    from behavior_driven_video_synthesis_amd.models.imagenet_pretrained import PerceptualVGG, vgg19
    meta, arr = load_golden("g3_losses")
    seed = meta["seed"]
    means = [seeded_randn("kl.m0", (3, 16, 4, 4), seed).cuda(), seeded_randn("kl.m1", (3, 16, 8, 8), seed).cuda()]
    logstds = [torch.sigmoid(seeded_randn("kl.l0", (3, 16, 4, 4), seed)).cuda(),
               torch.sigmoid(seeded_randn("kl.l1", (3, 16, 8, 8), seed)).cuda()]
    assert_close(compute_kl_with_prior(means, logstds), arr["kl"], rtol=1e-5, atol=1e-3, name="kl")
    assert_close(compute_kl_loss([means[0]], [logstds[0]]), arr["latent_kl"], rtol=1e-5, atol=1e-3, name="latent_kl")
    pv = PerceptualVGG(vgg19(seed=meta["vgg_seed"]), meta["loss_weights"]).cuda()
    t = synth_image("vgg.t", (2, 3, 32, 32), seed).cuda()
    p = synth_image("vgg.p", (2, 3, 32, 32), seed).cuda().requires_grad_(True)
    feats = pv(t)
    assert list(feats.keys()) == meta["tap_order"]
    for k, v in feats.items():
        d = v.double()
        assert_close(torch.stack([d.mean(), d.abs().mean(), d.std()]), arr[f"tap.{k}.stats"], rtol=1e-4, atol=1e-6,
                     name=k + ".stats")
        assert_close(v.flatten()[:64], arr[f"tap.{k}.head"], rtol=1e-3, atol=1e-4, name=k + ".head")
    ld = vgg_loss(pv, t, p)
    for k, v in ld.items():
        assert_close(v, arr["vggloss." + k], rtol=1e-4, atol=1e-6, name="vggloss." + k)
    torch.stack(list(ld.values()), 0).sum().backward()
    assert_close(p.grad, arr["vggloss.gp"], rtol=2e-3, atol=2e-6, name="vggloss.gp")


def test_discriminators_vs_golden():
    from behavior_driven_video_synthesis_amd.models.synth_discriminator import PartDiscriminator, PatchGANDiscriminator
    meta, arr = load_golden("g4_discriminators")
    seed = meta["seed"]
    pd = PartDiscriminator(n_scales=2, part_size=16)
    assert {k: list(v.shape) for k, v in pd.state_dict().items()} == meta["part_shapes"]
    pd.load_state_dict(synth_state_dict(meta["part_shapes"], seed))
    pd = pd.cuda()
    x = synth_image("pd.x", (2, 3, 18, 18), seed).cuda()
    assert_close(pd(x), arr["pd.out"], name="pd.out")

    pg = PatchGANDiscriminator(3, ndf=8, n_layers=3)
    assert {k: list(v.shape) for k, v in pg.state_dict().items()} == meta["patch_shapes"]
    pg.load_state_dict(synth_state_dict(meta["patch_shapes"], seed))
    pg = pg.cuda()
    x = synth_image("pg.x", (2, 3, 32, 32), seed).cuda().requires_grad_(True)
    out = pg(x)
    assert_close(out, arr["pg.out"], name="pg.out")
    (out * seeded_randn("pg.w", tuple(out.shape), seed).cuda()).sum().backward()
    assert_close(x.grad, arr["pg.gx"], rtol=2e-3, atol=1e-4, name="pg.gx")
    params = dict(pg.named_parameters())
    for k, s in meta["pg_grad_sums"].items():
        got = float(params[k].grad.double().abs().sum())
        assert abs(got - s[1]) <= 2e-3 * s[1] + 1e-4, (k, got, s[1])


def test_training_trajectory_vs_golden():
    """K fused-Adam steps of the train_fn loss assembly reproduce the reference's torch.optim.Adam trajectory."""
    from behavior_driven_video_synthesis_amd.models.vunets import VunetAlter
    from behavior_driven_video_synthesis_amd.models.imagenet_pretrained import PerceptualVGG, vgg19
    from behavior_driven_video_synthesis_amd.lib.losses import compute_kl_with_prior, vgg_loss
    from behavior_driven_video_synthesis_amd.optim import FusedAdam
    from oracle import vunet_oracle as O
    meta, arr = load_golden("g5_trajectory")
    seed, cfg = meta["seed"], meta["cfg"]
    net = VunetAlter(n_channels_x=3, **cfg)
    net.load_state_dict(synth_state_dict(meta["shapes"], seed))
    net = net.cuda().train()
    pv = PerceptualVGG(vgg19(seed=meta["vgg_seed"], width_div=meta["vgg_width_div"]), [1.0] * 6).cuda()
    opt = FusedAdam([{"params": list(getattr(net, n).parameters()), "name": n} for n in ["eu", "ed", "du", "dd"]],
                    lr=meta["lr0"], betas=tuple(meta["betas"]))
    gamma, lr = meta["gamma0"], meta["lr0"]
    for rec in meta["steps"]:
        it = rec["it"]
        x = synth_image(f"traj.x{it}", (2, 3, 32, 32), seed).cuda()
        c = synth_image(f"traj.c{it}", (2, 3, 32, 32), seed).cuda()
        eps = [seeded_randn(f"traj.{it}.eps{i}", s, seed).cuda() for i, s in enumerate([(2, 16, 4, 4), (2, 16, 8, 8)])]
        img, means, logstds, _ = net(x, c, eps)
        ld = vgg_loss(pv, x, img)
        ll = 1.0 * torch.sum(torch.stack([ld[k] for k in ld], dim=0))
        kl = compute_kl_with_prior(means, logstds)
        loss = ll + gamma * kl if it > meta["n_init_batches"] else ll
        for got, key in ((loss, "loss"), (ll, "ll"), (kl, "kl")):
            assert abs(float(got) - rec[key]) <= 5e-4 * abs(rec[key]) + 1e-5, (it, key, float(got), rec[key])
        opt.zero_grad()
        loss.backward()
        opt.step()
        gamma = O.update_gamma(gamma, meta["gamma_step"], meta["imax"], float(kl))
        lr = O.linear_var(it, 0, meta["total_steps"], meta["lr0"], 0, 0, meta["lr0"])
        for g in opt.param_groups:
            g["lr"] = lr
    sd = net.state_dict()
    assert_close(sd["dd.out_conv.conv.weight_v"], arr["final.dd.out_conv.conv.weight_v"], rtol=2e-3, atol=2e-5,
                 name="final weight")
    for k, s in meta["param_checksums"].items():
        got = float(sd[k].double().abs().sum())
        assert abs(got - s[1]) <= 5e-4 * s[1] + 1e-5, (k, got, s[1])


def test_full_width_vunet_vs_oracle():
    """The real Human3.6m layer widths (nf 32..128, 7 scales) at 128^2, bs 2: HIP vs the CPU oracle."""
    from behavior_driven_video_synthesis_amd.models.vunets import VunetAlter
    from oracle import vunet_oracle as O
    cfg = dict(spatial_size=128, bottleneck_factor=2, box_factor=2, n_scales=0, n_latent_scales=2,
               conv_layer_type="l1", nf_start=32, nf_max=128, subpixel_upsampling=True, dropout_prob=0.0)
    net = VunetAlter(n_channels_x=3, **cfg)
    shapes = {k: list(v.shape) for k, v in net.state_dict().items()}
    sd = synth_state_dict(shapes, 3)
    net.load_state_dict(sd)
    net = net.cuda().train()
    x, c = synth_image("fx", (2, 3, 128, 128), 3), synth_image("fc", (2, 3, 128, 128), 3)
    eps = [seeded_randn("fe0", (2, 128, 4, 4), 3), seeded_randn("fe1", (2, 128, 8, 8), 3)]
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    img_r, means_r, logstds_r, _ = O.vunet_alter_forward(sdr, cfg, x, c, eps)
    wgt = seeded_randn("fw", tuple(img_r.shape), 3)
    (img_r * wgt).sum().backward()
    img, means, logstds, _ = net(x.cuda(), c.cuda(), [e.cuda() for e in eps])
    scale = float(img_r.abs().max())
    assert_close(img, img_r, rtol=1e-4, atol=1e-4 * max(scale, 1.0), name="img")
    assert psnr(img, img_r, peak=2 * scale) >= 100.0
    (img * wgt.cuda()).sum().backward()
    for k, p in net.named_parameters():
        gr = sdr[k].grad
        if gr is None:
            continue
        tol = 2e-3 * float(gr.abs().max()) + 1e-6
        assert_close(p.grad, gr, rtol=2e-3, atol=tol, name=k)


@pytest.mark.parametrize("conv_layer_type", ["l2", "ln"])
def test_vunet_alter_conv_layer_variants_vs_oracle(conv_layer_type):
    """conv_layer_type l2 (L2NormConv2d) and anything else (LayerNormConv2d): models/vunets.py:445-453, SURVEY a15."""
    from behavior_driven_video_synthesis_amd.models.vunets import VunetAlter
    from oracle import vunet_oracle as O
    cfg = dict(spatial_size=32, bottleneck_factor=2, box_factor=2, n_scales=0, n_latent_scales=2,
               conv_layer_type=conv_layer_type, nf_start=8, nf_max=16, subpixel_upsampling=True, dropout_prob=0.0)
    net = VunetAlter(init_fn=lambda: False, **cfg)
    shapes = {k: list(v.shape) for k, v in net.state_dict().items()}
    sd = synth_state_dict(shapes, 8)
    net.load_state_dict(sd)
    net = net.cuda().train()
    x, c = synth_image("vx", (2, 3, 32, 32), 8), synth_image("vc", (2, 3, 32, 32), 8)
    eps = [seeded_randn("ve0", (2, 16, 4, 4), 8), seeded_randn("ve1", (2, 16, 8, 8), 8)]
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    img_r, means_r, logstds_r, _ = O.vunet_alter_forward(sdr, cfg, x, c, eps)
    wgt = seeded_randn("vw", tuple(img_r.shape), 8)
    (img_r * wgt).sum().backward()
    img, means, logstds, _ = net(x.cuda(), c.cuda(), [e.cuda() for e in eps])
    scale = max(float(img_r.abs().max()), 1.0)
    assert_close(img, img_r, rtol=2e-4, atol=2e-4 * scale, name="img")
    assert_close(means[1], means_r[1], rtol=2e-4, atol=2e-4 * scale, name="mean1")
    (img * wgt.cuda()).sum().backward()
    gmax = max(float(v.grad.abs().max()) for v in sdr.values() if v.grad is not None)
    for k, p in net.named_parameters():
        gr = sdr[k].grad
        if gr is None:
            continue
        # a conv bias in front of InstanceNorm has a mathematically zero gradient: both sides are rounding noise,
        # so the absolute tolerance is tied to the largest gradient of the model
        tol = 3e-3 * float(gr.abs().max()) + 1e-4 * gmax + 1e-6
        assert_close(p.grad, gr, rtol=3e-3, atol=tol, name=k)


def test_r1_penalty_double_backward_vs_golden():
    """compute_grad2 (models/synth_discriminator.py:244-256): gradient of the logits w.r.t. the input, squared, and
    its own backward -- a double backward through the HIP conv kernels (forward <-> dgrad <-> wgrad primitives)."""
    from behavior_driven_video_synthesis_amd.models.synth_discriminator import PartDiscriminator, compute_grad2
    meta, arr = load_golden("g4_discriminators")
    seed = meta["seed"]
    pd = PartDiscriminator(n_scales=2, part_size=16)
    pd.load_state_dict(synth_state_dict(meta["part_shapes"], seed))
    pd = pd.cuda()
    x = synth_image("pd.x", (2, 3, 18, 18), seed).cuda().requires_grad_(True)
    out = pd(x)
    reg = compute_grad2(out, x).mean()
    assert_close(reg, arr["pd.reg"], rtol=1e-3, atol=1e-6, name="reg")
    (out.sum() + 10.0 * reg).backward()
    assert_close(x.grad, arr["pd.gx"], rtol=3e-3, atol=2e-4, name="gx")
    params = dict(pd.named_parameters())
    for k, s in meta["pd_grad_sums"].items():
        got = float(params[k].grad.double().abs().sum())
        assert abs(got - s[1]) <= 3e-3 * s[1] + 1e-4, (k, got, s[1])


def test_disc_trainer_step_vs_golden():
    """DiscTrainer.train_disc with the R1 penalty + get_genloss with gradient-ratio weighting (:139-206)."""
    from behavior_driven_video_synthesis_amd.models.synth_discriminator import DiscTrainer
    meta, arr = load_golden("g4_discriminators")
    seed = meta["seed"]

    class Gen(torch.nn.Module):      # the test's stand-in generator tail (a torch 1x1 conv), as in make_golden.py
        def __init__(self):
            super().__init__()
            self.last = torch.nn.Conv2d(3, 3, 1)
    gen = Gen()
    gen.load_state_dict(synth_state_dict({k: list(v.shape) for k, v in gen.state_dict().items()}, seed))
    gen = gen.cuda()
    tr = DiscTrainer(gen, {"pd_scales": 2, "adam_beta": (0.5, 0.9), "save_intervall": 10}, spatial_size=64,
                     grad_pen=True, lambda_gp=10, grad_weighting=True)
    tr.disc.load_state_dict(synth_state_dict(meta["part_shapes"], seed))
    tr.init_training([torch.device("cuda:0")], lr=1e-3)
    real = synth_image("dt.real", (2, 3, 18, 18), seed).cuda()
    zin = synth_image("dt.z", (2, 3, 18, 18), seed).cuda()
    out = tr.train_disc(real, gen.last(zin).detach())
    for k, v in meta["train_disc"].items():
        assert abs(out[k] - v) <= 2e-3 * abs(v) + 1e-5, (k, out[k], v)
    sd = tr.disc.state_dict()
    for k, s in meta["disc_after"].items():
        got = float(sd[k].double().abs().sum())
        assert abs(got - s[1]) <= 1e-3 * s[1] + 1e-5, (k, got, s[1])
    fake = gen.last(zin)
    gl, wgt = tr.get_genloss(fake, (fake - real).abs().mean(), gen.last.weight)
    assert_close(gl, arr["dt.gen_loss"], rtol=1e-3, atol=1e-5, name="gen_loss")
    assert_close(wgt, arr["dt.loss_weight"], rtol=2e-2, atol=1e-4, name="loss_weight")


@pytest.mark.parametrize("tag", ["h36m256", "market128"])
def test_full_size_vunet_vs_reference_fixture_and_oracle(tag):
    """BASELINE config 2 (Human3.6m, 256^2, nf 32..128, 7 scales) and config 1 (Market, 128^2, 30-channel 64x64
    appearance input, bottleneck_factor 1, box_factor 1; README.md:103-110) at bs 2: the HIP path against (i) the
    statistics / crop / gradient slices recorded from the imported reference (G6) and (ii) the CPU oracle on every
    output element (PSNR >= 100 dB) and every parameter gradient."""
    from behavior_driven_video_synthesis_amd.models.vunets import VunetAlter
    from oracle import vunet_oracle as O
    from test_oracle_golden import check_full_size
    meta, arr = load_golden("g6_" + tag)
    seed, cfg, ncx = meta["seed"], meta["cfg"], meta["n_channels_x"]
    net = VunetAlter(n_channels_x=ncx, **cfg)
    shapes = {k: list(v.shape) for k, v in net.state_dict().items()}
    assert len(shapes) == meta["n_keys"] and sum(p.numel() for p in net.parameters()) == meta["n_params"]
    sd = synth_state_dict(shapes, seed)
    net.load_state_dict(sd)
    net = net.cuda().train()
    x, c = synth_image(tag + ".x", tuple(meta["x"]), seed), synth_image(tag + ".c", tuple(meta["c"]), seed)
    eps = [seeded_randn(f"{tag}.eps{i}", tuple(s), seed) for i, s in enumerate(meta["eps_shapes"])]
    wgt = seeded_randn(tag + ".w", (x.shape[0], 3, cfg["spatial_size"], cfg["spatial_size"]), seed)
    img, means, logstds, _ = net(x.cuda(), c.cuda(), [e.cuda() for e in eps])
    (img * wgt.cuda()).sum().backward()
    grads = {k: (None if p.grad is None else p.grad.detach().cpu()) for k, p in net.named_parameters()}
    check_full_size(meta, arr, img.detach().cpu(), [m.detach().cpu() for m in means],
                    [l.detach().cpu() for l in logstds], grads)
    # every element against the oracle
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    img_r, means_r, logstds_r, _ = O.vunet_alter_forward(sdr, cfg, x, c, eps, n_channels_x=ncx)
    (img_r * wgt).sum().backward()
    scale = float(img_r.abs().max())
    assert_close(img, img_r, rtol=1e-4, atol=1e-4 * max(scale, 1.0), name="img")
    assert psnr(img, img_r, peak=2 * scale) >= 100.0
    for i in range(len(means)):
        assert_close(means[i], means_r[i], rtol=1e-4, atol=1e-4, name=f"mean{i}")
        assert_close(logstds[i], logstds_r[i], rtol=1e-4, atol=1e-4, name=f"logstd{i}")
    for k, p in net.named_parameters():
        gr = sdr[k].grad
        if gr is None:
            continue
        tol = 2e-3 * float(gr.abs().max()) + 1e-6
        assert_close(p.grad, gr, rtol=2e-3, atol=tol, name=k)


def test_pretrained_directory_written_by_the_reference_restores_and_renders():
    """a checkpoint directory in the authors' layout (config.yaml + reg_ckpt*.pth written by the reference's
    own classes, tests/golden/make_golden.py g8_pretrained_dir) goes through the ``--pretrained_model`` flow and the restored
    model renders what the reference model rendered (``transfer``, same inputs and posterior noise)."""
    import os
    from conftest import GOLDEN, load_golden
    from hip_parity_utils import assert_close
    from synth import seeded_randn, synth_image
    from behavior_driven_video_synthesis_amd.experiments.checkpoint import load_pretrained
    meta, arr = load_golden("g8_pretrained_outputs")
    seed = meta["seed"]
    tr, _ = load_pretrained(os.path.join(GOLDEN, "g8_pretrained"), device="cuda:0", vgg_synthetic=True, vgg_width_div=8)
    tr.vunet.eval()
    x, c = synth_image("pre.tx", (2, 3, 32, 32), seed).cuda(), synth_image("pre.tc", (2, 3, 32, 32), seed).cuda()
    eps = [seeded_randn(f"pre.t.eps{i}", s, seed).cuda() for i, s in enumerate([(2, 8, 4, 4), (2, 8, 8, 8)])]
    with torch.no_grad():
        img = tr.vunet.transfer(x, c, eps)
    assert_close(img, arr["transfer"], name="transfer")


def test_relu_masks_applied_by_the_gradient_producer_change_nothing():
    """ops.enable_relu_premask: inside the VGG stack the [x > 0] mask of a conv + ReLU layer's backward is applied by
    whoever PRODUCES that layer's output gradient (the next layer's data-gradient epilogue, max-pool, the L1 tap), and the
    layer then runs the plain data-gradient kernel instead of masking while it stages.  Same gradient (the fp16 scheme
    scales by the maxima of the masked instead of the unmasked tensor: fp32 rounding apart), different kernels."""
    from behavior_driven_video_synthesis_amd import ops
    from behavior_driven_video_synthesis_amd.lib.losses import vgg_loss
    from behavior_driven_video_synthesis_amd.models.imagenet_pretrained import PerceptualVGG, vgg19
    pv = PerceptualVGG(vgg19(seed=3, width_div=2), [1.0, 0.5, 1.5, 1.0, 2.0, 1.0]).cuda()
    g = torch.Generator().manual_seed(11)
    target = (torch.rand(2, 3, 128, 128, generator=g) * 2 - 1).cuda()
    pred0 = (torch.rand(2, 3, 128, 128, generator=g) * 2 - 1).cuda()

    def run(on):
        ops.enable_relu_premask(on)
        try:
            pred = pred0.clone().requires_grad_(True)
            ops.profile_start()
            losses = vgg_loss(pv, target, pred)
            torch.stack([v.sum() for v in losses.values()]).sum().backward()
            fam = ops.profile_stop(by_kernel=True)
            return pred.grad.clone(), {k: float(v.sum()) for k, v in losses.items()}, fam
        finally:
            ops.enable_relu_premask(True)
    g_on, l_on, k_on = run(True)
    g_off, l_off, k_off = run(False)
    assert l_on == l_off
    scale = float(g_off.abs().max())
    assert float((g_on - g_off).abs().max()) <= 2e-5 * scale, (float((g_on - g_off).abs().max()), scale)
    masked = lambda fam: sum(v["n"] for k, v in fam.items() if ", 1, 4, " in k)     # data gradient masking in its staging
    assert masked(k_off) >= 1 and masked(k_on) == 0, (sorted(k_on), sorted(k_off))


def test_vgg_loss_and_gradient_vs_oracle_with_the_fused_backward_paths():
    """lib/losses.py:81-119 on a 128x128 batch with half-width VGG19: large enough for the split-fp16 row-tiled kernels,
    the pools and every fused backward path of round 3 (L1 tap + pool in one node, gradient pass-through into the L1
    kernel, ReLU masks applied by the gradient's producer) -- loss terms and d loss / d pred against the CPU oracle.

    The gradient of an L1 loss on ReLU / max-pool features is discontinuous in the features (sign(p - t), the pool's
    argmax), so fp32 rounding moves it by far more than 1e-6: the oracle evaluated in float32 is itself 5e-3 (relative
    L2) from the oracle evaluated in float64 (tools/vgg_grad_cmp.py).  The reference here is therefore the float64
    oracle, and the bar is "at least as close to it as the float32 oracle is" -- measured: 3.5e-4."""
    from behavior_driven_video_synthesis_amd import ops
    from behavior_driven_video_synthesis_amd.lib.losses import vgg_loss
    from behavior_driven_video_synthesis_amd.models.imagenet_pretrained import PerceptualVGG, vgg19
    from oracle import vunet_oracle as O
    weights = [1.0, 0.5, 1.5, 1.0, 2.0, 1.0]
    pv = PerceptualVGG(vgg19(seed=77, width_div=2), weights).cuda()
    vsd = O.make_synthetic_vgg19(seed=77, width_div=2)
    target = synth_image("vl.t", (2, 3, 128, 128), 5)
    pred0 = synth_image("vl.p", (2, 3, 128, 128), 6)

    def oracle(dtype):
        p_ = pred0.clone().to(dtype).requires_grad_(True)
        ld_ = O.vgg_loss({k: v.to(dtype) for k, v in vsd.items()}, weights, target.to(dtype), p_)
        torch.stack([v.sum() for v in ld_.values()]).sum().backward()
        return ld_, p_.grad.double()
    ld64, g64 = oracle(torch.float64)
    _, g32 = oracle(torch.float32)
    p = pred0.cuda().requires_grad_(True)
    ops.profile_start()
    ld = vgg_loss(pv, target.cuda(), p)
    torch.stack([v.sum() for v in ld.values()]).sum().backward()
    fam = ops.profile_stop(by_kernel=True)
    assert list(ld) == list(ld64)
    for k in ld:
        assert_close(ld[k].cpu(), ld64[k].detach().float().reshape(ld[k].shape), rtol=1e-4, atol=1e-6, name="vgg_loss." + k)
    got = p.grad.double().cpu()
    rel = float((got - g64).norm() / g64.norm())
    rel32 = float((g32 - g64).norm() / g64.norm())
    assert rel <= 2e-3 and rel <= rel32, (rel, rel32)
    assert float((got - g64).abs().max()) <= 1e-2 * float(g64.abs().max())
    assert any(k.startswith("conv_h2_kernel<") for k in fam), sorted(fam)      # the fp16 row-tiled kernels really ran


@pytest.mark.parametrize("planes", [True, False])
def test_full_width_vgg19_loss_and_gradient_at_256_vs_float64_oracle(planes):
    """(``planes``: the stack on pre-split fp16 planes from relu1_1 up -- csrc/conv_p2.hip, the default -- or on fp32 NCHW
    tensors through the per-layer h2 kernels; same arithmetic, same bars.)

    VERDICT r3 weak #2: the perceptual loss at the BENCHMARK shape -- full-width VGG19 (64 .. 512 channels), 256x256,
    batch 2: every layer form the bs-16 step runs (64-channel 256^2 rows, the 512-channel 32^2 layers, the 16-wide
    conv5_x form) -- the six loss terms to 1e-4 and d loss / d pred against the CPU oracle evaluated in float64.

    Measured (tools/vgg_grad_cmp.py 256 1 2, profiles/r04_vgg_grad_cmp_256.txt), relative L2 distance to the float64 oracle:
    the float32 CPU oracle 2.6e-3, this path (split fp16, fp32 accumulation on the matrix cores) 3.3e-3, the fp32-input MFMA
    kernels 3.8e-3; features: relu1_2 2.0e-7 / 2.1e-7 / 3.3e-7 ... relu5_2 5.8e-7 / 9.8e-7 / 1.9e-6.  Every float32
    evaluation scatters around float64 at this level -- the gradient of an L1 loss on ReLU / max-pool features is
    discontinuous (sign(p - t), the pool's argmax), so feature errors of 1e-6 flip signs -- and at K up to 4608 the CPU's
    blocked summation is the most accurate of the three, as it was not at the half-width 128^2 size.  The bar is therefore
    not "closer to float64 than the float32 oracle" but "within 1.5x of the float32 oracle's own distance and below 5e-3"."""
    from behavior_driven_video_synthesis_amd import ops
    from behavior_driven_video_synthesis_amd.lib.losses import vgg_loss
    from behavior_driven_video_synthesis_amd.models.imagenet_pretrained import PerceptualVGG, vgg19
    from behavior_driven_video_synthesis_amd.models import imagenet_pretrained as ip
    from oracle import vunet_oracle as O
    weights = [1.0, 0.5, 1.5, 1.0, 2.0, 1.0]
    pv = PerceptualVGG(vgg19(seed=78, width_div=1), weights).cuda()
    vsd = O.make_synthetic_vgg19(seed=78, width_div=1)
    target = synth_image("vl256.t", (2, 3, 256, 256), 5)
    pred0 = synth_image("vl256.p", (2, 3, 256, 256), 6)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, 32))    # (PyTorch-CPU thrashes on this path with hundreds of threads)
    try:
        def oracle(dtype):
            p_ = pred0.clone().to(dtype).requires_grad_(True)
            ld_ = O.vgg_loss({k: v.to(dtype) for k, v in vsd.items()}, weights, target.to(dtype), p_)
            torch.stack([v.sum() for v in ld_.values()]).sum().backward()
            return ld_, p_.grad.double()
        ld64, g64 = oracle(torch.float64)
        _, g32 = oracle(torch.float32)
    finally:
        torch.set_num_threads(threads)
    p = pred0.cuda().requires_grad_(True)
    ip.enable_p2(planes)
    try:
        ops.profile_start()
        ld = vgg_loss(pv, target.cuda(), p)
        torch.stack([v.sum() for v in ld.values()]).sum().backward()
        fam = ops.profile_stop(by_kernel=True)
    finally:
        ip.enable_p2(True)
    assert list(ld) == list(ld64)
    for k in ld:
        assert_close(ld[k].cpu(), ld64[k].detach().float().reshape(ld[k].shape), rtol=1e-4, atol=1e-6, name="vgg_loss256." + k)
    got = p.grad.double().cpu()
    rel = float((got - g64).norm() / g64.norm())
    rel32 = float((g32 - g64).norm() / g64.norm())
    assert rel <= 5e-3 and rel <= 1.5 * rel32, (rel, rel32)
    assert float((got - g64).abs().max()) <= 3e-2 * float(g64.abs().max())   # (2.1e-2 measured, for the float32 oracle too)
    names = sorted(fam)
    if planes:   # every 3x3 layer past conv1_1, forward and data gradient, on the p2 kernels (32-wide and 16-wide tiles)
        assert any(k.startswith("conv_p2") and "32" in k for k in names) and any(k.startswith("conv_p2") and "16" in k for k in names), names
        assert not any(k.startswith("conv_h2_kernel") for k in names), names
        return
    assert any(k.startswith("conv_h2_kernel<2, 2, 0, 0") for k in names), names      # the step's dominant kernel
    # the 512-channel layers: one-row-tile forms at 32^2, and conv5_x (16 x 16 maps: the 16-wide form at bs 16, the
    # small-map kernel at this batch)
    assert any(k.startswith("conv_h2_kernel<2, 1, 0, 0") for k in names), names
    assert any(k.endswith(", 16>") or k.startswith("conv_h2_small_kernel<") for k in names), names
