"""Behaviour front half of BASELINE config 5 on the GPU (csrc/seq.hip through the C ABI): the flow in both directions and
the behaviour net's encoder / decoder against the fixtures the reference's own classes wrote (tests/golden/g9_behavior.npz)
and against the pinned oracle (oracle/behavior_oracle.py) at sizes up to the reference configuration's."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from synth import seeded_randn, synth_behavior_state

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _no_grad():
    """The behaviour path is inference only and says so when autograd is recording (last test of this file)."""
    with torch.no_grad():
        yield

TOL = dict(rtol=2e-4, atol=2e-5)   # fp32 chains of 20 - 60 layers / 50 recurrent steps, different summation order
FLOW_TOL = dict(rtol=1e-3, atol=1e-4)   # the small FIXTURE flows (synthetic O(1) scale nets: every tanh saturated, a pass multiplies
#                                         by e^(+-1) per half-coupling and by 1/scale per block): rounding differences of the MLP
#                                         sums grow with that condition number; measured 3.7e-4 on one element of 320, 1e-6
#                                         typical.  Flows conditioned like a trained one are held to the float64 oracle instead,
#                                         relative to the CPU float32 oracle's own distance from it (the width tests below)


def close(a, b, **kw):
    tol = dict(TOL)
    tol.update(kw)
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a, b, **tol)


def _sd(info, arr, prefix, seed):
    stored = {k[len(prefix) + 4:]: torch.from_numpy(v) for k, v in arr.items() if k.startswith(prefix + ".sd.")}
    return synth_behavior_state(info["shapes"], seed, stored)


def _flow(kw, sd):
    from behavior_driven_video_synthesis_amd.models.flow.simple_flow import UnsupervisedTransformer2
    flow = UnsupervisedTransformer2(**kw)
    flow.load_state_dict(sd)          # strict: the key set is the reference's
    return flow.cuda()


def _net(kw, sd):
    from behavior_driven_video_synthesis_amd.models.pose_behavior_rnn import ResidualBehaviorNet
    net = ResidualBehaviorNet(**kw)
    net.load_state_dict(sd)
    return net.cuda()


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("tag", ["even", "odd"])
def test_flow_vs_reference_fixture(tag, graph):
    meta, arr = load_golden("g9_behavior")
    seed, info = meta["seed"], meta["cases"][f"flow_{tag}"]
    flow = _flow(info["kw"], _sd(info, arr, f"flow_{tag}", seed))
    flow.flow.engine().graph.enabled = graph
    chan, bsz = info["kw"]["flow_in_channels"], info["batch"]
    x = seeded_randn(f"flow.{tag}.x", (bsz, chan), seed).cuda()
    z = seeded_randn(f"flow.{tag}.z", (bsz, chan), seed).cuda()
    for _ in range(2):                                   # second call: the replayed graph
        out, logdet = flow(x)
        rev = flow.reverse(z)
        assert out.shape == (bsz, chan, 1, 1) and rev.shape == (bsz, chan, 1, 1)
        close(out.reshape(bsz, chan), arr[f"flow_{tag}.forward"], **FLOW_TOL)
        close(logdet, arr[f"flow_{tag}.logdet"], **FLOW_TOL)
        close(rev.reshape(bsz, chan), arr[f"flow_{tag}.reverse"], **FLOW_TOL)
    if chan % 2 == 0:
        close(flow.reverse(out).reshape(bsz, chan), x, rtol=1e-3, atol=1e-4)
    close(flow.sample((bsz, chan)).shape, (bsz, chan))


@pytest.mark.parametrize("tag", ["plain", "nin"])
def test_behavior_net_vs_reference_fixture(tag):
    meta, arr = load_golden("g9_behavior")
    seed, info = meta["seed"], meta["cases"][f"net_{tag}"]
    net = _net(info["kw"], _sd(info, arr, f"net_{tag}", seed))
    bsz, t_in, length, n_kps, hid = info["batch"], info["t_in"], info["len"], info["kw"]["n_kps"], info["kw"]["dim_hidden_b"]
    x1 = (0.5 * seeded_randn(f"net.{tag}.x1", (bsz, t_in, n_kps), seed)).cuda()
    x2 = (0.5 * seeded_randn(f"net.{tag}.x2", (bsz, t_in, n_kps), seed)).cuda()
    b_given = seeded_randn(f"net.{tag}.b", (bsz, hid), seed).cuda()
    for _ in range(2):
        xs, cs, zs, b_ret = net.generate_seq(b_given, x2, len=length, start_frame=t_in - 1)
        close(xs, arr[f"net_{tag}.gen_xs"])
        close(cs, arr[f"net_{tag}.gen_cs"])
        assert zs == [] and b_ret is b_given
    eps = seeded_randn(f"net.{tag}.eps0", (bsz, hid), seed).cuda()
    xs, cs, _, b, mu, logstd, pre = net(x1, x2, length, start_frame=2, eps=eps)
    for name, v in dict(xs=xs, cs=cs, b=b, mu=mu, logstd=logstd, pre=pre).items():
        close(v, arr[f"net_{tag}.{name}"])
    noise = seeded_randn(f"net.{tag}.prior.eps0", (bsz, hid), seed).cuda()
    xs, _, _, b, *_ = net(x1, x2, length, start_frame=0, sample=True, eps=noise)
    close(b, arr[f"net_{tag}.b_prior"])
    close(xs, arr[f"net_{tag}.xs_prior"])
    # without an injected eps the module draws its own: shapes and finiteness only
    xs, cs, _, b, mu, logstd, pre = net(x1, x2, length)
    assert xs.shape == (bsz, length, n_kps) and torch.isfinite(xs).all() and not torch.equal(b, mu)


def _random_flow(chan, mid, depth, n_flows, seed, s_gain=1.0):
    """``s_gain`` < 1 shrinks the scale nets' last layer: with the recipe's O(1) pre-activations every tanh saturates and a
    pass multiplies by e^(+-1) per half-coupling -- a condition number no trained flow has, and one fp32 cannot carry."""
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


@pytest.mark.parametrize("bsz", [1, 16, 17, 48, 64, 100])
def test_flow_vs_oracle_over_batch_sizes(bsz):
    """Every batch-tile count of the kernel (1..4 tiles of 16 rows) and the chunking above 64 rows."""
    from oracle import behavior_oracle as B
    flow, sd = _random_flow(96, 160, 2, 2, 5)
    x = seeded_randn("bs.x", (bsz, 96), 5)
    z = seeded_randn("bs.z", (bsz, 96), 5)
    out, logdet = flow(x.cuda())
    o_ref, l_ref = B.flow_forward(sd, x)
    close(out.reshape(bsz, 96), o_ref, **FLOW_TOL)
    close(logdet, l_ref, **FLOW_TOL)
    close(flow.reverse(z.cuda()).reshape(bsz, 96), B.flow_reverse(sd, z), **FLOW_TOL)


def test_flow_at_the_reference_width_vs_oracle_and_round_trip():
    """config/behavior_net.yaml: 1024 channels, 2048 hidden, depth 2 (4 of the 15 blocks: 670 MB of weights), 16 rows as
    experiments/behavior_net.py:1173 samples them; graph replay must equal the eager issue bit for bit."""
    from oracle import behavior_oracle as B
    flow, sd = _random_flow(1024, 2048, 2, 4, 9, s_gain=0.1)
    eng = flow.flow.engine()
    z = seeded_randn("full.z", (16, 1024), 9)
    x_ref = B.flow_reverse(sd, z)
    x_f64 = B.flow_reverse({k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}, z.double())
    eng.graph.enabled = False
    x_eager = flow.reverse(z.cuda())
    eng.graph.enabled = True
    x_graph = [flow.reverse(z.cuda()) for _ in range(3)][-1]
    assert torch.equal(x_eager, x_graph)
    # against the float64 oracle, next to the CPU float32 oracle's own distance from it
    e_hip = float((x_graph.reshape(16, 1024).cpu().double() - x_f64).abs().max() / x_f64.abs().max())
    e_cpu = float((x_ref.double() - x_f64).abs().max() / x_f64.abs().max())
    print(f"flow reverse at 1024 / 2048 x 4 blocks: max err / max|x|  HIP {e_hip:.2e}  CPU fp32 {e_cpu:.2e}")
    assert e_hip <= max(3.0 * e_cpu, 2e-6), (e_hip, e_cpu)
    close(x_graph.reshape(16, 1024), x_ref, **FLOW_TOL)
    back, logdet = flow(x_graph)
    close(back.reshape(16, 1024), z, rtol=1e-3, atol=2e-4)
    _, l_ref = B.flow_forward(sd, x_ref)
    close(logdet, l_ref, rtol=1e-3, atol=1e-2)


def test_decoder_roll_out_at_the_reference_size_vs_oracle():
    """dim_hidden_b 1024, 51 pose dimensions, 50 steps (config/behavior_net.yaml), 16 rows; graph == eager."""
    from oracle import behavior_oracle as B
    from behavior_driven_video_synthesis_amd.models.pose_behavior_rnn import ResidualBehaviorNet
    torch.manual_seed(3)
    net = ResidualBehaviorNet(51, information_bottleneck=True, decoder_arch="lstm", linear_in_decoder=False, dim_hidden_b=1024)
    sd = synth_behavior_state({k: list(v.shape) for k, v in net.state_dict().items()}, 3, {})
    net.load_state_dict(sd)
    net = net.cuda()
    b = seeded_randn("dec.b", (16, 1024), 3)
    seq = 0.5 * seeded_randn("dec.x", (16, 50, 51), 3)
    xs_ref, cs_ref = B.generate_seq(sd, b, seq, 50, 49)
    xs_f64, _ = B.generate_seq({k: v.double() for k, v in sd.items()}, b.double(), seq.double(), 50, 49)
    eng = net.engine()
    eng.graph.enabled = False
    xs_e, cs_e, _, _ = net.generate_seq(b.cuda(), seq.cuda(), len=50, start_frame=49)
    eng.graph.enabled = True
    xs_g = [net.generate_seq(b.cuda(), seq.cuda(), len=50, start_frame=49)[0] for _ in range(3)][-1]
    assert torch.equal(xs_e, xs_g)
    # 50 recurrent steps: against the float64 oracle, next to the CPU float32 oracle's own distance from it
    e_hip = float((xs_g.cpu().double() - xs_f64).abs().max() / xs_f64.abs().max())
    e_cpu = float((xs_ref.double() - xs_f64).abs().max() / xs_f64.abs().max())
    print(f"decoder roll-out, 50 steps at 1024 hidden: max err / max|x|  HIP {e_hip:.2e}  CPU fp32 {e_cpu:.2e}")
    assert e_hip <= max(3.0 * e_cpu, 2e-6), (e_hip, e_cpu)
    close(xs_g, xs_ref, rtol=1e-3, atol=5e-4)
    close(cs_e, cs_ref, rtol=1e-3, atol=5e-4)
    eps = seeded_randn("dec.eps", (16, 1024), 3)
    b2, mu, logstd, pre = net.infer_b(seq.cuda(), False, eps=eps.cuda())
    rb, rmu, rls, rpre = B.infer_b(sd, seq, eps)
    for got, ref in ((b2, rb), (mu, rmu), (logstd, rls), (pre, rpre)):
        close(got, ref, rtol=1e-3, atol=1e-4)


def test_actnorm_data_dependent_initialisation():
    """lib/modules.py:270-290, :303-305: the first forward of a fresh flow sets loc / scale from the batch, block by block."""
    from behavior_driven_video_synthesis_amd.lib.modules import ActNorm
    from oracle import behavior_oracle as B
    x = seeded_randn("an.x", (24, 40), 2) * 3.0 + 1.5
    an = ActNorm(40, logdet=True).cuda()
    h, logdet = an(x.cuda())
    mean, std = x.mean(0), x.std(0)
    close(an.loc.reshape(-1), -mean, rtol=1e-5, atol=1e-6)
    close(an.scale.reshape(-1), 1.0 / (std + 1e-6), rtol=1e-5, atol=1e-6)
    assert int(an.initialized.item()) == 1
    close(h, (x - mean) / (std + 1e-6) * 1.0, rtol=1e-4, atol=1e-5)
    close(logdet, torch.log(1.0 / (std + 1e-6)).sum().expand(24), rtol=1e-5, atol=1e-4)
    close(an.reverse(h), x, rtol=1e-4, atol=1e-5)
    # a fresh flow: after the first forward every block is initialised and its input statistics were normalised
    from behavior_driven_video_synthesis_amd.models.flow.simple_flow import UnsupervisedTransformer2
    torch.manual_seed(4)
    flow = UnsupervisedTransformer2(flow_in_channels=32, flow_mid_channels=64, flow_hidden_depth=1, n_flows=3).cuda()
    xb = (seeded_randn("an.flow", (32, 32), 2) * 2.0 - 0.5).cuda()
    out, logdet = flow(xb)
    assert all(int(blk.norm_layer.initialized.item()) == 1 for blk in flow.flow.sub_layers)
    sd = {k: v.detach().cpu() for k, v in flow.state_dict().items()}
    o_ref, l_ref = B.flow_forward(sd, xb.cpu())          # the oracle with the parameters the init produced
    close(out.reshape(32, 32), o_ref, **FLOW_TOL)
    close(logdet, l_ref, rtol=1e-4, atol=1e-3)
    close(sd["flow.sub_layers.0.norm_layer.loc"].reshape(-1), -xb.cpu().mean(0), rtol=1e-5, atol=1e-6)
    out2, _ = flow(xb)                                     # second call: the fused, graph-replayed pass
    close(out2, out, rtol=1e-5, atol=1e-6)


def test_stand_alone_mlp_and_the_cpu_refusal():
    from behavior_driven_video_synthesis_amd.lib.modules import BasicFullyConnectedNet
    from oracle import behavior_oracle as B
    for tanh in (False, True):
        net = BasicFullyConnectedNet(dim=17, depth=2, hidden_dim=48, use_tanh=tanh, out_dim=9)
        sd = synth_behavior_state({k: list(v.shape) for k, v in net.state_dict().items()}, 6, {})
        net.load_state_dict(sd)
        x = seeded_randn("mlp.x", (70, 17), 6)
        close(net.cuda()(x.cuda()), B.fully_connected({"p." + k: v for k, v in sd.items()}, "p", x, tanh))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(x)


def test_a_module_left_on_the_cpu_or_cast_is_refused():
    """The kernels take parameters as raw pointers: a module on another device than its input, or cast away from fp32, raises
    before any launch (a CUDA input with host pointers would otherwise fault the GPU)."""
    from behavior_driven_video_synthesis_amd.lib.modules import ActNorm, BasicFullyConnectedNet
    from behavior_driven_video_synthesis_amd.models.flow.simple_flow import UnsupervisedTransformer2
    from behavior_driven_video_synthesis_amd.models.pose_behavior_rnn import ResidualBehaviorNet
    z = torch.randn(4, 32, device="cuda")
    flow = UnsupervisedTransformer2(flow_in_channels=32, flow_mid_channels=48, flow_hidden_depth=1, n_flows=1)
    for blk in flow.flow.sub_layers:
        blk.norm_layer.initialized.fill_(1)
    with pytest.raises(RuntimeError, match="is on cpu"):
        flow.reverse(z)
    with pytest.raises(RuntimeError, match="is on cpu"):
        flow(z)
    with pytest.raises(RuntimeError, match="float64"):
        flow.cuda().double().reverse(z)
    assert flow.float().reverse(z).shape == (4, 32, 1, 1)
    with pytest.raises(RuntimeError, match="is on cpu"):
        BasicFullyConnectedNet(dim=32, depth=1, hidden_dim=32)(z)
    with pytest.raises(RuntimeError, match="is on cpu"):
        ActNorm(32)(z)
    with pytest.raises(RuntimeError, match="is on cpu"):
        ActNorm(32).reverse(z)
    net = ResidualBehaviorNet(51, information_bottleneck=True, decoder_arch="lstm", dim_hidden_b=64)
    with pytest.raises(RuntimeError, match="is on cpu"):
        net.generate_seq(torch.randn(2, 64, device="cuda"), torch.randn(2, 3, 51, device="cuda"), len=2, start_frame=0)
    with pytest.raises(RuntimeError, match="is on cpu"):
        net.infer_b(torch.randn(2, 3, 51, device="cuda"), False)
    with pytest.raises(RuntimeError, match="float16"):
        net.cuda().half().infer_b(torch.randn(2, 3, 51, device="cuda"), False)


@pytest.mark.parametrize("shape", [(96, 160, 2, 2, 5), (33, 48, 1, 2, 4), (1024, 2048, 2, 2, 16)])
def test_tile_major_weights_and_activations_change_nothing(shape):
    """include/vunet_seq_tiled.h: the flow's inference plans read tile-major COPIES of the weight images and hand the hidden
    activations to each other tile-major (every wave load 1 KB contiguous); the arithmetic and its order are unchanged, so both
    directions, logdet included, are bit-identical to the row-major path -- padded widths, an odd channel count, the reference
    width; eager and replayed."""
    chan, mid, depth, n_flows, bsz = shape
    flow, _ = _random_flow(chan, mid, depth, n_flows, 31, s_gain=0.1)
    z = seeded_randn("tm.z", (bsz, chan), 31).cuda()
    x = seeded_randn("tm.x", (bsz, chan), 31).cuda()
    eng = flow.flow.engine()
    assert eng.TILED
    outs = {}
    for tiled in (True, False):
        eng.TILED = tiled
        eng._packed_for = None
        eng._image_sig = None
        for graph in (False, True):
            eng.graph.enabled = graph
            rev = [flow.reverse(z) for _ in range(2)][-1]
            fwd, ld = [flow(x) for _ in range(2)][-1]
            outs[(tiled, graph)] = (rev.clone(), fwd.clone(), ld.clone())
        assert all((h.wt is not None) == tiled for blk in eng.blocks for h in blk["halves"])
    ref = outs[(False, False)]
    for key, got in outs.items():
        for a, b in zip(got, ref):
            assert torch.equal(a, b), key


def test_full_flow_of_the_reference_configuration_vs_oracle():
    """config/behavior_net.yaml's WHOLE flow -- 1024 channels, 2048 hidden, depth 2, all 15 blocks (629 M
    parameters, 2.5 GB) -- in both directions at 16 rows (experiments/behavior_net.py:1173 samples that many), where the
    e^(+-s) conditioning compounds over 30 half-couplings: against the float64 oracle, next to the CPU float32 oracle's own
    distance from it, and the round trip.  Graph replay equals eager issue bit for bit."""
    from oracle import behavior_oracle as B
    flow, sd = _random_flow(1024, 2048, 2, 15, 21, s_gain=0.02)
    # conditioned like a trained flow (every block near the identity, activations O(1) through all 15): the recipe's O(1)
    # translation nets and ActNorm scales of 0.7 - 1 compound to a map that float32 cannot carry on the CPU either (measured
    # with s_gain 0.1 alone: CPU float32 0.81 of max|x| from float64 on the reverse pass)
    for k in list(sd):
        if ".coupling.t." in k and ".main.6." in k:
            sd[k] = sd[k] * 0.2
        elif k.endswith("norm_layer.scale"):
            sd[k] = 1.0 + 0.05 * seeded_randn(k, tuple(sd[k].shape), 21)
        elif k.endswith("norm_layer.loc"):
            sd[k] = 0.05 * seeded_randn(k, tuple(sd[k].shape), 21)
    flow.load_state_dict(sd)
    z = seeded_randn("full.z", (16, 1024), 21)
    x = seeded_randn("full.x", (16, 1024), 21)
    eng = flow.flow.engine()
    eng.graph.enabled = False
    rev_e = flow.reverse(z.cuda()).reshape(16, 1024)
    fwd_e, ld_e = flow(x.cuda())
    eng.graph.enabled = True
    rev = [flow.reverse(z.cuda()).reshape(16, 1024) for _ in range(2)][-1]
    fwd, ld = [flow(x.cuda()) for _ in range(2)][-1]
    assert torch.equal(rev, rev_e) and torch.equal(fwd, fwd_e) and torch.equal(ld, ld_e)
    sd64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}
    r32, r64 = B.flow_reverse(sd, z), B.flow_reverse(sd64, z.double())
    f32, l32 = B.flow_forward(sd, x)
    f64, l64 = B.flow_forward(sd64, x.double())

    def dist(a, ref):
        return float((a.detach().cpu().double() - ref).abs().max() / ref.abs().max())
    rows = {"reverse": (dist(rev, r64), dist(r32, r64)), "forward": (dist(fwd.reshape(16, 1024), f64), dist(f32, f64)),
            "logdet": (dist(ld, l64), dist(l32, l64))}
    for name, (e_hip, e_cpu) in rows.items():
        print(f"15-block flow, {name}: max err / max|.| vs float64  HIP {e_hip:.2e}  CPU fp32 {e_cpu:.2e}")
        assert e_hip <= max(3.0 * e_cpu, 2e-6), (name, e_hip, e_cpu)
    back = flow.reverse(fwd).reshape(16, 1024)
    rt = dist(back, x.double())
    print(f"15-block flow, reverse(forward(x)) - x: {rt:.2e} of max|x|")
    assert rt <= 1e-5      # measured 7.3e-7


@pytest.mark.parametrize("stats_dtype", ["float32", "float64"])
@pytest.mark.parametrize("dims", [51, 96])
def test_pose_projection_vs_the_reference_numpy_chain(dims, stats_dtype):
    """unNormalizeData + apply_affine_transform + camera_projection + joint rescale (data/data_conversions_3d.py:178-211, :588-605,
    :892-912, :1139-1140) in one launch, against the numpy restatement with the same dtypes."""
    from oracle import behavior_oracle as B
    from behavior_driven_video_synthesis_amd.render import PoseCamera
    rng = np.random.RandomState(11)
    ignore = sorted(rng.choice(dims, size=dims - 45, replace=False).tolist()) if dims == 96 else [3, 4, 5]
    use = [i for i in range(dims) if i not in ignore]
    mean = (rng.randn(dims) * 300.0).astype(stats_dtype)
    std = (50.0 + 200.0 * rng.rand(dims)).astype(stats_dtype)
    ang = 0.4
    rot = np.array([[np.cos(ang), 0.0, np.sin(ang)], [0.0, 1.0, 0.0], [-np.sin(ang), 0.0, np.cos(ang)]])
    ext = np.concatenate([rot, np.array([[50.0], [-120.0], [5200.0]])], axis=1)
    intr = (1145.0, 512.5, 1143.8, 515.4)
    x = rng.randn(37, len(use)).astype(np.float32)
    ref = B.poses_to_keypoints(x, mean, std, ignore, ext, intr, (1000, 1002), 256)
    cam = PoseCamera(mean, std, use, ext, intr, (1000, 1002), 256)
    got = cam.project(torch.from_numpy(x).cuda())
    assert got.shape == (37, dims // 3, 2)
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=2e-6, atol=2e-4)   # fp32 storage of pixel coordinates ~ 256


@pytest.mark.parametrize("tag", ["h36m_f32", "h36m_f64", "plain"])
def test_pose_projection_vs_the_reference_fixture(tag):
    """``seq_pose_project_kernel`` vs what the reference's own numpy functions produced (tests/golden/g9b_projection.npz:
    unNormalizeData, apply_affine_transform, camera_projection, joint rescale)."""
    from behavior_driven_video_synthesis_amd.render import PoseCamera
    meta, arr = load_golden("g9b_projection")
    c = meta["cases"][tag]
    use = [i for i in range(c["dims"]) if i not in c["ignore"]]
    cam = PoseCamera(arr[f"{tag}.mean"], arr[f"{tag}.std"], use, arr[f"{tag}.ext"], tuple(c["intrinsics"]), tuple(c["image_size"]),
                     c["spatial_size"])
    got = cam.project(torch.from_numpy(arr[f"{tag}.x"]).cuda())
    want = arr[f"{tag}.kps"]
    assert got.shape == want.shape
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-6, atol=2e-4)   # fp32 storage of pixel coordinates ~ 256


def test_behavior_video_end_to_end_vs_the_pieces():
    """BASELINE config 5 in one call: flow sample -> decoder roll-out -> projection -> raster -> VunetAlter.transfer; against the
    oracle's poses / keypoints for the same noise and the (separately tested) renderer on the oracle's keypoints."""
    from oracle import behavior_oracle as B
    from behavior_driven_video_synthesis_amd.models.pose_behavior_rnn import ResidualBehaviorNet
    from behavior_driven_video_synthesis_amd.models.vunets import VunetAlter
    from behavior_driven_video_synthesis_amd.render import PoseCamera, behavior_video, render_sequence
    from synth import synth_image, synth_state_dict
    flow, fsd = _random_flow(64, 96, 1, 2, 13, s_gain=0.3)
    torch.manual_seed(13)
    net = ResidualBehaviorNet(51, information_bottleneck=True, decoder_arch="lstm", linear_in_decoder=False, dim_hidden_b=64)
    nsd = synth_behavior_state({k: list(v.shape) for k, v in net.state_dict().items()}, 13, {})
    nsd["decoder.n_out.weight"] = nsd["decoder.n_out.weight"] * 0.05      # small steps: the figure stays in the image
    net.load_state_dict(nsd)
    net = net.cuda()
    cfg = dict(spatial_size=64, bottleneck_factor=2, box_factor=2, n_scales=0, n_latent_scales=2, conv_layer_type="l1", nf_start=8,
               nf_max=16, subpixel_upsampling=True, dropout_prob=0.0)
    vunet = VunetAlter(**cfg)
    vunet.load_state_dict(synth_state_dict({k: list(v.shape) for k, v in vunet.state_dict().items()}, 13))
    vunet = vunet.cuda().eval()
    # a plausible standing figure: joints around (0, 0, 0) +- 600 mm, camera 5 m away
    rng = np.random.RandomState(5)
    mean = (rng.randn(51) * 250.0).astype(np.float32)
    std = (40.0 + 40.0 * rng.rand(51)).astype(np.float32)
    ext = np.concatenate([np.eye(3), np.array([[0.0], [0.0], [5000.0]])], axis=1)
    cam = PoseCamera(mean, std, list(range(51)), ext, (1145.0, 500.0, 1145.0, 500.0), (1000, 1000), 64)
    bsz, t_in, length = 2, 5, 6
    start = 0.5 * seeded_randn("e2e.start", (bsz, t_in, 51), 13)
    z = seeded_randn("e2e.z", (bsz, 64), 13)
    app = synth_image("e2e.app", (1, 3, 64, 64), 13)
    frames, poses, kps = behavior_video(flow, net, vunet, app.cuda(), start.cuda(), length, cam, z=z.cuda(), dtype="f32",
                                        share_appearance=False)
    assert frames.shape == (bsz, length, 64, 64, 3) and frames.dtype == torch.uint8
    b_ref = B.flow_reverse(fsd, z)
    poses_ref, _ = B.generate_seq(nsd, b_ref, start, length, t_in - 1)
    close(poses, poses_ref, rtol=1e-3, atol=1e-4)
    kps_ref = B.poses_to_keypoints(poses_ref.reshape(bsz * length, 51).numpy(), mean, std, [], ext, (1145.0, 500.0, 1145.0, 500.0),
                                   (1000, 1000), 64).reshape(bsz, length, 17, 2)
    np.testing.assert_allclose(kps.cpu().numpy(), kps_ref, rtol=1e-4, atol=2e-2)
    assert float(kps.min()) > -64 and float(kps.max()) < 128          # the figure is on (or near) the canvas
    for i in range(bsz):   # the same renderer on the oracle's keypoints: identical unless a keypoint sits on a rounding edge
        rgb, stick = render_sequence(vunet, app.cuda(), torch.from_numpy(kps_ref[i]).float().cuda(), spatial_size=64, dtype="f32")
        same = (rgb.int() - frames[i].int()).abs() <= 1
        assert float(same.float().mean()) > 0.995
        assert float(stick.abs().sum()) > 0   # something was drawn


def test_recordings_and_their_buffers_are_capped_together():
    """A caller that keeps changing the roll-out length does not accumulate hipGraphs; an evicted recording takes its input /
    output buffers with it and a later call with the same key records afresh (same results)."""
    from behavior_driven_video_synthesis_amd import seq as seq_mod
    from behavior_driven_video_synthesis_amd.models.pose_behavior_rnn import ResidualBehaviorNet
    torch.manual_seed(8)
    net = ResidualBehaviorNet(51, information_bottleneck=True, decoder_arch="lstm", dim_hidden_b=64).cuda()
    eng = net.engine()
    b = torch.randn(3, 64, device="cuda")
    x = 0.5 * torch.randn(3, 4, 51, device="cuda")
    first = net.generate_seq(b, x, len=2, start_frame=0)[0].clone()
    cap = seq_mod._GraphCache.MAX_GRAPHS
    for length in range(3, 3 + cap + 4):
        net.generate_seq(b, x, len=length, start_frame=0)
    assert len(eng.graph.graphs) <= cap
    assert all(len(p["io"]) <= cap for p in eng._plans.values())
    assert all(k in eng.graph.graphs for p in eng._plans.values() for k in p["io"])
    again = net.generate_seq(b, x, len=2, start_frame=0)[0]
    assert torch.equal(first, again)


def test_a_recorded_call_trains_or_is_refused():
    """With autograd recording and trainable parameters, the flow's forward direction and the behaviour net's forward come back
    WITH a graph (csrc/seq_train.hip, csrc/seq_bptt.hip: tests/test_hip_seq_train.py); the pieces that have no backward -- the
    flow's reverse direction, the stand-alone building blocks -- raise instead of handing back tensors without one."""
    from behavior_driven_video_synthesis_amd.lib.modules import ActNorm, BasicFullyConnectedNet
    flow, _ = _random_flow(32, 48, 1, 1, 3)
    z = torch.randn(2, 32, device="cuda")
    with torch.enable_grad():
        with pytest.raises(RuntimeError, match="inference only"):
            flow.reverse(z)
        out, logdet = flow(z)
        assert out.requires_grad and logdet.requires_grad
        with pytest.raises(RuntimeError, match="inference only"):
            BasicFullyConnectedNet(dim=16, depth=1, hidden_dim=32).cuda()(z[:, :16])
        with pytest.raises(RuntimeError, match="inference only"):
            ActNorm(32).cuda().reverse(z)
        for p in flow.parameters():
            p.requires_grad_(False)
        assert flow.reverse(z).shape == (2, 32, 1, 1)     # frozen parameters: nothing to record


@pytest.mark.parametrize("bsz,hid,nin", [(64, 128, False), (40, 64, True), (33, 1024, False)])
def test_roll_outs_with_h_handed_on_as_tiles_vs_the_rows_and_the_oracle(bsz, hid, nin, monkeypatch):
    """More than 32 rows: the LSTM steps of the decoder roll-out and of the encoder read h from the tile-major copy the step
    before left (``vunet_seq_lstm_gates_tiled_h``) -- bit-identical with ``VUNET_SEQ_LSTM_TILED_H=0`` (everything from the rows),
    and both close to the oracle."""
    from oracle import behavior_oracle as B
    from behavior_driven_video_synthesis_amd.models.pose_behavior_rnn import ResidualBehaviorNet
    outs = {}
    for tiled in ("1", "0"):
        monkeypatch.setenv("VUNET_SEQ_LSTM_TILED_H", tiled)
        net = ResidualBehaviorNet(51, information_bottleneck=True, decoder_arch="lstm", linear_in_decoder=nin, dim_hidden_b=hid)
        sd = synth_behavior_state({k: list(v.shape) for k, v in net.state_dict().items()}, 19, {})
        net.load_state_dict(sd)
        net = net.cuda()
        b = seeded_randn("th.b", (bsz, hid), 19)
        x = 0.5 * seeded_randn("th.x", (bsz, 6, 51), 19)
        eps = seeded_randn("th.eps", (bsz, hid), 19)
        xs, cs, _, _ = net.generate_seq(b.cuda(), x.cuda(), len=7, start_frame=2)
        enc = net.infer_b(x.cuda(), False, eps=eps.cuda())
        plan = net.engine()._plans[bsz]
        assert (plan["ht"] is not None) == (tiled == "1" and hid % 32 == 0 and net.engine().hoff % 32 == 0)
        outs[tiled] = (xs, cs) + tuple(enc)
    for a, b_ in zip(outs["1"], outs["0"]):
        assert torch.equal(a, b_)
    xs_ref, cs_ref = B.generate_seq(sd, b, x, 7, 2)
    close(outs["1"][0], xs_ref, rtol=1e-3, atol=5e-4)
    close(outs["1"][1], cs_ref, rtol=1e-3, atol=5e-4)
    for g, r in zip(outs["1"][2:], B.infer_b(sd, x, eps)):
        close(g, r, rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("bsz,t_in,length,start", [(1, 1, 1, 0), (65, 3, 4, -1), (16, 2, 9, 1)])
def test_decoder_and_encoder_edge_shapes_vs_oracle(bsz, t_in, length, start):
    """One row / one frame / one step; more than 64 rows (two chunks); a negative start frame as the reference indexes it."""
    from oracle import behavior_oracle as B
    from behavior_driven_video_synthesis_amd.models.pose_behavior_rnn import ResidualBehaviorNet
    torch.manual_seed(12)
    net = ResidualBehaviorNet(51, information_bottleneck=True, decoder_arch="lstm", linear_in_decoder=True, dim_hidden_b=96)
    sd = synth_behavior_state({k: list(v.shape) for k, v in net.state_dict().items()}, 12, {})
    net.load_state_dict(sd)
    net = net.cuda()
    b = seeded_randn("edge.b", (bsz, 96), 12)
    x = 0.5 * seeded_randn("edge.x", (bsz, t_in, 51), 12)
    xs, cs, _, _ = net.generate_seq(b.cuda(), x.cuda(), len=length, start_frame=start)
    xs_ref, cs_ref = B.generate_seq(sd, b, x, length, start)
    close(xs, xs_ref, rtol=5e-4, atol=5e-5)
    close(cs, cs_ref, rtol=5e-4, atol=5e-5)
    eps = seeded_randn("edge.eps", (bsz, 96), 12)
    got = net.infer_b(x.cuda(), False, eps=eps.cuda())
    ref = B.infer_b(sd, x, eps)
    for g, r in zip(got, ref):
        close(g, r, rtol=5e-4, atol=5e-5)
