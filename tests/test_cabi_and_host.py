"""CPU-only checks: the C-ABI library loads and exports every symbol include/vunet_hip.h declares, the
host-side mirror has the reference's construction surface / state-dict layout, schedules match the
oracle, and the product path refuses CPU tensors (there is no fallback)."""
import os
import re

import pytest
import torch

from conftest import ROOT, load_golden


def test_library_exports_every_declared_symbol():
    from behavior_driven_video_synthesis_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    lib = _lib.lib()
    names = _lib.declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), n
    assert lib.vunet_abi_version() == 12


def test_header_is_plain_c_abi():
    txt = open(os.path.join(ROOT, "include", "vunet_hip.h")).read()
    assert 'extern "C"' in txt
    assert "torch" not in re.sub(r"/\*.*?\*/", "", txt, flags=re.S).lower()
    assert "at::" not in txt and "hipStream_t stream" not in txt  # streams cross the ABI as void*


def test_ctypes_structs_match_header_field_order():
    from behavior_driven_video_synthesis_amd import ops, seq
    txt = open(os.path.join(ROOT, "include", "vunet_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    for cname, cls in (("vunet_conv_desc", ops.ConvDesc), ("vunet_wgrad_desc", ops.WgradDesc),
                       ("vunet_wn_desc", ops.WnDesc), ("vunet_p2_desc", ops.P2Desc),
                       ("vunet_seq_linear_desc", seq.SeqLinearDesc), ("vunet_seq_coupling_desc", seq.SeqCouplingDesc),
                       ("vunet_seq_lstm_desc", seq.SeqLstmDesc)):
        body = re.search(r"typedef struct " + cname + r" \{(.*?)\} " + cname, txt, flags=re.S).group(1)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if decl:
                typ, names = decl.split(None, 1)
                fields += [(n.strip(), typ) for n in names.split(",")]
        assert [f[0] for f in cls._fields_] == [f[0] for f in fields], cname
        cmap = {"int32_t": "c_int", "uint32_t": "c_uint", "float": "c_float", "int64_t": "c_long"}
        for (n, ct), (_, typ) in zip(cls._fields_, fields):
            assert ct.__name__.startswith(cmap[typ]), (cname, n)


def test_behavior_modules_keep_the_reference_state_dict_layout_and_refuse_the_cpu():
    """The flow / behaviour-net mirrors (BASELINE config 5's front half): key names, order and shapes of the reference's
    ``UnsupervisedTransformer2`` / ``ResidualBehaviorNet`` as the g9 fixture recorded them; no CPU path."""
    import pytest
    import torch
    from behavior_driven_video_synthesis_amd.models.flow.simple_flow import UnsupervisedTransformer2
    from behavior_driven_video_synthesis_amd.models.pose_behavior_rnn import ResidualBehaviorNet
    meta, _ = load_golden("g9_behavior")
    for tag in ("flow_even", "flow_odd"):
        info = meta["cases"][tag]
        flow = UnsupervisedTransformer2(**info["kw"])
        sd = flow.state_dict()
        assert list(sd.keys()) == list(info["shapes"].keys())
        assert {k: list(v.shape) for k, v in sd.items()} == info["shapes"]
        assert sd["flow.sub_layers.0.shuffle.forward_shuffle_idx"].dtype == torch.int64
        perm = sd["flow.sub_layers.0.shuffle.forward_shuffle_idx"]
        assert torch.equal(perm[sd["flow.sub_layers.0.shuffle.backward_shuffle_idx"]], torch.arange(perm.numel()))
        assert flow.get_last_layer() is flow.flow.sub_layers[-1].coupling.t[-1].linears()[-1].weight
    for tag in ("net_plain", "net_nin"):
        info = meta["cases"][tag]
        net = ResidualBehaviorNet(**info["kw"])
        sd = net.state_dict()
        assert list(sd.keys()) == list(info["shapes"].keys())
        assert {k: list(v.shape) for k, v in sd.items()} == info["shapes"]
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        flow.reverse(torch.zeros(2, flow.in_channels))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net.generate_seq(torch.zeros(2, 64), torch.zeros(2, 3, 51), len=2, start_frame=0)
    with pytest.raises(NotImplementedError):
        ResidualBehaviorNet(51, decoder_arch="gru", dim_hidden_b=64)     # (the reference's GRU decoder cannot run either)


def test_state_dict_layout_matches_reference():
    from behavior_driven_video_synthesis_amd.models.vunets import Regressor, VunetAlter, VunetOrg
    from behavior_driven_video_synthesis_amd.models.synth_discriminator import PartDiscriminator, PatchGANDiscriminator
    for tag, cls in (("g2_alter", VunetAlter), ("g2_alter_box", VunetAlter), ("g2_org", VunetOrg)):
        meta, _ = load_golden(tag)
        net = cls(n_channels_x=meta["n_channels_x"], **meta["cfg"])  # unknown cfg keys must be ignored
        sd = {k: list(v.shape) for k, v in net.state_dict().items()}
        assert list(sd.keys()) == list(meta["shapes"].keys())
        assert sd == meta["shapes"]
        assert hasattr(net, "eu") and hasattr(net, "ed") and hasattr(net, "du") and hasattr(net, "dd")
    meta, _ = load_golden("g4_discriminators")
    assert {k: list(v.shape) for k, v in PartDiscriminator(2, 16).state_dict().items()} == meta["part_shapes"]
    assert {k: list(v.shape) for k, v in PatchGANDiscriminator(3, 8, 3).state_dict().items()} == meta["patch_shapes"]
    meta, _ = load_golden("g2_regressor")
    reg = Regressor(n_out=34, n_latent_scales=2, nf_max=16, latent_widths=[8, 4], linear_width_factor=1)
    assert {k: list(v.shape) for k, v in reg.state_dict().items()} == meta["shapes"]


def test_full_size_model_has_reference_parameter_count():
    from behavior_driven_video_synthesis_amd.models.vunets import VunetAlter
    cfg = dict(spatial_size=256, bottleneck_factor=2, box_factor=2, n_scales=0, n_latent_scales=2,
               conv_layer_type="l1", nf_start=32, nf_max=128, subpixel_upsampling=True)
    net = VunetAlter(**cfg)
    assert sum(p.numel() for p in net.parameters()) == 14601516  # SURVEY 8a, Human3.6m config
    assert len(net.state_dict()) == 495 and net.n_scales == 7


def test_cpu_tensors_raise_no_fallback():
    from behavior_driven_video_synthesis_amd.lib.modules import NormConv2d
    from behavior_driven_video_synthesis_amd.lib.losses import compute_kl_with_prior
    with pytest.raises(RuntimeError):
        NormConv2d(4, 4, 3, 1, 1)(torch.zeros(1, 4, 8, 8))
    with pytest.raises(RuntimeError):
        compute_kl_with_prior([torch.zeros(2, 4, 2, 2)], [torch.zeros(2, 4, 2, 2)])


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "behavior_driven_video_synthesis_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+.*oracle", src, flags=re.M), os.path.join(dp, f)


def test_schedules_match_oracle():
    from oracle import vunet_oracle as O
    from behavior_driven_video_synthesis_amd.lib.utils import linear_var
    for it in (0, 1, 17, 99, 100, 150):
        assert abs(float(linear_var(it, 0, 100, 5e-4, 0, 0, 5e-4)) - O.linear_var(it, 0, 100, 5e-4, 0, 0, 5e-4)) < 1e-15
    assert O.update_gamma(0.0, 1e-5, 1000.0, 3000.0) == pytest.approx(0.02)
    assert O.update_gamma(0.001, 1e-5, 1000.0, 10.0) == 0.0


def test_dropout_hash_restatement_is_stable():
    """The CPU restatement of the conv-prologue dropout hash used by the parity tests (known answers)."""
    import numpy as np
    from hip_parity_utils import dropout_keep_mask, hash_u32
    assert [int(v) for v in hash_u32(np.array([0, 1, 2, 0xFFFFFFFF], dtype=np.uint64))] == \
        [0, 1753845952, 3507691905, 1734902346]
    m = dropout_keep_mask((4, 1000), 0.25, 12345)
    assert 0.70 < float(m.mean()) < 0.80


def test_fused_adam_reads_and_writes_torch_adam_checkpoints():
    """Checkpoint interop (experiments/shape_and_pose_net.py:474-482, experiments/experiment.py:57-66): the
    optimizer entry of a reference checkpoint is a torch.optim.Adam state dict with extra group keys."""
    from behavior_driven_video_synthesis_amd.optim import FusedAdam
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.Linear(4, 2))
    ref = torch.optim.Adam([{"params": net[0].parameters(), "name": "eu"}, {"params": net[1].parameters(), "name": "dd"}],
                           lr=5e-4, betas=(0.5, 0.9))
    for _ in range(3):
        ref.zero_grad()
        net(torch.randn(7, 5)).square().mean().backward()
        ref.step()
    for pg in ref.param_groups:
        pg["gamma"] = 0.25          # the reference smuggles gamma through the param groups (:507-512)
    sd = ref.state_dict()
    net2 = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.Linear(4, 2))
    net2.load_state_dict(net.state_dict())
    opt = FusedAdam([{"params": list(net2[0].parameters()), "name": "eu"},
                     {"params": list(net2[1].parameters()), "name": "dd"}], lr=1e-3)
    opt.load_state_dict(sd)
    assert [b.step for b in opt.buckets] == [3, 3]
    assert opt.param_groups[0]["lr"] == 5e-4 and opt.param_groups[1]["gamma"] == 0.25
    assert tuple(opt.param_groups[0]["betas"]) == (0.5, 0.9)
    out = opt.state_dict()
    assert set(out["state"].keys()) == set(sd["state"].keys())
    for k in sd["state"]:
        assert torch.allclose(out["state"][k]["exp_avg"], sd["state"][k]["exp_avg"])
        assert torch.allclose(out["state"][k]["exp_avg_sq"], sd["state"][k]["exp_avg_sq"])
        assert int(out["state"][k]["step"]) == 3
    assert [g["name"] for g in out["param_groups"]] == ["eu", "dd"] and out["param_groups"][0]["params"] == [0, 1]


def test_checkpoint_file_selection_matches_reference_rule(tmp_path):
    """experiments/experiment.py:44-60: among *.pth files containing the key, the largest trailing _<number> wins."""
    from behavior_driven_video_synthesis_amd.experiments.checkpoint import latest_checkpoint, load_ckpt
    d = str(tmp_path)
    for name, val in [("reg_ckpt_model_900.pth", 1), ("reg_ckpt_model_10000.pth", 2), ("reg_ckpt_model_2000.pth", 3),
                      ("regressor_model_99999.pth", 4)]:
        torch.save({"model": {"w": torch.tensor(float(val))}, "optimizer": {"state": {}, "param_groups": []}},
                   os.path.join(d, name))
    assert os.path.basename(latest_checkpoint(d, "reg_ckpt")) == "reg_ckpt_model_10000.pth"
    mod, opt = load_ckpt(d, "reg_ckpt")
    assert float(mod["w"]) == 2.0 and opt == {"state": {}, "param_groups": []}
    assert load_ckpt(d, "nothing_like_this") == (None, None)


def test_kernel_selection_of_the_gather_entry_point():
    """vunet_conv2d_gather_variant is host logic (no GPU): which kernel each kind of layer of the path is routed to."""
    import ctypes
    from behavior_driven_video_synthesis_amd import _lib, ops

    def variant(has_aux=0, **kw):
        base = dict(N=16, C1=64, C2=0, Hs=128, Ws=128, M=64, m_off=0, Mpad=64, Ho=128, Wo=128, KH=3, KW=3, stride=1, pad=1,
                    mode=0, in_act=0, in_slope=0.0, drop_p=0.0, drop_seed=0, out_act=0, d2s=0)
        base.update(kw)
        buf = ctypes.create_string_buffer(96)
        assert _lib.lib().vunet_conv2d_gather_variant(ctypes.byref(ops.ConvDesc(**base)), has_aux, buf, 96) == 0
        return buf.value.decode()

    # VGG19 stack: plain 3x3 layers on the tallest LDS tile that leaves two workgroups per CU
    assert variant(C1=256, M=256, Mpad=256, Hs=64, Ws=64, Ho=64, Wo=64) == "conv_tiled_kernel<2, 4, 4, 0, 0, 32, 1>"
    assert variant(C1=512, M=512, Mpad=512, Hs=32, Ws=32, Ho=32, Wo=32) == "conv_tiled_kernel<2, 2, 8, 0, 0, 32, 1>"
    assert variant(C1=512, M=512, Mpad=512, Hs=16, Ws=16, Ho=16, Wo=16) == "conv_tiled_kernel<2, 1, 8, 0, 0, 16, 1>"
    # residual-block conv with ELU + dropout prologue; its data gradient (act'(aux) epilogue) keeps the 4-row tile
    assert variant(in_act=ops.ACT_ELU, drop_p=0.05).startswith("conv_tiled_kernel<2, ")
    assert variant(mode=1, has_aux=1) == "conv_tiled_kernel<2, 1, 8, 1, 0, 32, 1>"
    # 1x1 nin layers stream; 8x8 maps share the K loop between 16 waves; stride-2 data gradient runs per output parity
    assert variant(KH=1, KW=1, pad=0, C1=32, M=32, Mpad=32, Hs=256, Ws=256, Ho=256, Wo=256,
                   in_act=ops.ACT_ELU) == "conv_1x1_kernel<1, 0, 1>"
    assert variant(C1=128, M=128, Mpad=128, Hs=8, Ws=8, Ho=8, Wo=8) == "conv_gather_splitk_kernel<16, 0, 3>"
    assert variant(mode=1, stride=2, C1=128, M=64, Hs=64, Ws=64, Ho=128, Wo=128) == "conv_gather_kernel<phase x4>"
    # three channels on one side: VALU kernels
    assert variant(C1=32, M=3, Mpad=32, Hs=256, Ws=256, Ho=256, Wo=256) == "conv_thin_mv_kernel<0, 4>"
    assert variant(C1=3, M=32, Mpad=32, KH=1, KW=1, pad=0, Hs=256, Ws=256, Ho=256, Wo=256) == "conv_thin_k_kernel<1, 3>"
    assert variant(C1=3, M=64, Mpad=64, Hs=256, Ws=256, Ho=256, Wo=256, out_act=ops.ACT_RELU) == "conv_thin_kv_kernel<3, 4>"


def test_p2_dispatch_and_the_vgg19_program_are_host_logic():
    """The pre-split VGG19 path (csrc/conv_p2.hip, models/imagenet_pretrained._P2Engine) without a GPU: which geometries
    vunet_p2_conv covers and which workgroup form it picks, the weight image's size, and the program the engine derives from
    the ``features`` stack (first layer, conv / pool steps, taps) -- incl. the stacks it must refuse."""
    import ctypes
    import torch
    from behavior_driven_video_synthesis_amd import _lib, ops
    from behavior_driven_video_synthesis_amd.models.imagenet_pretrained import PerceptualVGG, _P2Engine, vgg19
    lib = _lib.lib()

    def variant(n, c, h, w, m):
        d = ops.P2Desc(n, c, h, w, m, 1)
        if lib.vunet_p2_conv_supported(ctypes.byref(d)) != 1:
            return None
        buf = ctypes.create_string_buffer(96)
        assert lib.vunet_p2_conv_variant(ctypes.byref(d), buf, 96) == 0
        return buf.value.decode()
    # the bs-16 256^2 stack: eight waves in antiphase wherever 128-channel workgroups fill the chip, four waves otherwise
    assert variant(16, 64, 256, 256, 64) == "conv_p2_kernel<4, 32, 1>"          # conv1_2
    assert variant(16, 128, 128, 128, 128) == "conv_p2a_kernel<32>"              # conv2_2
    assert variant(16, 512, 32, 32, 512) == "conv_p2a_kernel<32>"                # conv4_x
    assert variant(16, 512, 32, 32, 256) == "conv_p2_kernel<4, 32, 1>"           # conv4_1's data gradient: 128 wide workgroups would be 128
    assert variant(16, 512, 16, 16, 512) == "conv_p2_kernel<4, 16, 1>"           # conv5_x
    assert variant(2, 48, 32, 32, 64) is None and variant(2, 64, 12, 32, 64) is None and variant(2, 64, 16, 24, 64) is None
    assert variant(2, 64, 16, 32, 96) is None
    assert lib.vunet_p2_weight_image_bytes(512, 512, 0) == 512 * 512 * 9 * 4 and lib.vunet_p2_weight_image_bytes(64, 3, 0) == 0
    assert lib.vunet_p2_weight_image_bytes(64, 128, 1) == 64 * 128 * 9 * 4
    pv = PerceptualVGG(vgg19(synthetic=True, pretrained=True), [1.0] * 6)
    first, steps = _P2Engine.build_program(list(pv.vgg_layers._modules.items()), 31, pv.target_layers)
    assert tuple(first.weight.shape) == (64, 3, 3, 3)
    assert [s_[0] for s_ in steps] == ["conv", "pool", "conv", "conv", "pool"] + ["conv"] * 4 + ["pool"] + ["conv"] * 4 + ["pool", "conv", "conv"]
    assert [s_[2] for s_ in steps if s_[0] == "conv" and s_[2]] == ["relu1_2", "relu2_2", "relu3_2", "relu4_2", "relu5_2"]
    # a stack whose first layer is tapped, or with a bare conv (no ReLU behind it), is not p2-shaped: the fp32-tensor path serves it
    assert _P2Engine.build_program(list(pv.vgg_layers._modules.items()), 31, {"1": "relu1_1", "31": "relu5_2"}) is None
    assert _P2Engine.build_program(list(pv.vgg_layers._modules.items())[:1], 31, pv.target_layers) is None
    # no engine on the CPU (and none for a half-width stack's 32-channel layers): features_for_loss falls back to forward()
    assert pv._p2_engine(torch.zeros(1, 3, 64, 64)) is None


def test_dropin_runs_an_unchanged_main_against_a_checkout(tmp_path):
    """``python -m behavior_driven_video_synthesis_amd.dropin main.py``: the reference's import lines resolve to the
    MI355X classes, names this package does not define fall through to the checkout's own files, and ``lib`` / ``models``
    stay the checkout's packages (other sub-modules untouched).  A miniature checkout stands in for the reference."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    co = tmp_path / "checkout"
    (co / "lib").mkdir(parents=True)
    (co / "models").mkdir()
    (co / "lib" / "modules.py").write_text("class NormConv2d:\n    ORIGIN = 'checkout'\nclass GINActNorm:\n    ORIGIN = 'checkout'\n"
                                            "class ActNorm:\n    ORIGIN = 'checkout'\nclass BasicFullyConnectedNet:\n    ORIGIN = 'checkout'\n")
    (co / "lib" / "utils.py").write_text("def helper():\n    return 'checkout utils'\n")
    (co / "lib" / "losses.py").write_text("from lib.utils import helper\nclass FlowLoss:\n    ORIGIN = helper()\n")
    (co / "models" / "vunets.py").write_text("class VunetAlter:\n    ORIGIN = 'checkout'\n")
    (co / "models" / "flow").mkdir()
    (co / "models" / "flow" / "__init__.py").write_text("")
    (co / "models" / "flow" / "simple_flow.py").write_text("class UnsupervisedTransformer2:\n    ORIGIN = 'checkout'\nclass SupervisedTransformer:\n    ORIGIN = 'checkout'\n")
    (co / "models" / "pose_behavior_rnn.py").write_text("class ResidualBehaviorNet:\n    ORIGIN = 'checkout'\nclass MTVAE:\n    ORIGIN = 'checkout'\n")
    (co / "main.py").write_text(
        "from models.vunets import VunetAlter, Regressor\n"
        "from lib.modules import NormConv2d, GINActNorm, ActNorm, BasicFullyConnectedNet\n"
        "from lib.losses import vgg_loss, compute_kl_with_prior, FlowLoss\n"
        "from lib.utils import helper\n"
        "from models.imagenet_pretrained import PerceptualVGG\n"
        "from models.synth_discriminator import DiscTrainer\n"
        "from models.flow.simple_flow import UnsupervisedTransformer2, SupervisedTransformer\n"
        "from models.pose_behavior_rnn import ResidualBehaviorNet, MTVAE\n"
        "import sys\n"
        "print('ARGS', sys.argv[1:])\n"
        "print('BEHAVIOR', UnsupervisedTransformer2.__module__, ResidualBehaviorNet.__module__, SupervisedTransformer.ORIGIN, MTVAE.ORIGIN)\n"
        "print('VUNET', VunetAlter.__module__)\n"
        "print('NORMCONV', NormConv2d.__module__)\n"
        "print('ACTNORM', GINActNorm.ORIGIN)\n"
        "print('FLOWBLOCKS', ActNorm.ORIGIN, BasicFullyConnectedNet.ORIGIN)\n"
        "print('FLOWLOSS', FlowLoss.ORIGIN)\n"
        "print('UTILS', helper())\n"
        "print('VGGLOSS', vgg_loss.__module__, PerceptualVGG.__module__, DiscTrainer.__module__)\n")
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-m", "behavior_driven_video_synthesis_amd.dropin", "main.py", "--config", "x.yaml"],
                       cwd=str(co), env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    out = dict(line.split(" ", 1) for line in r.stdout.strip().splitlines())
    pkg = "behavior_driven_video_synthesis_amd"
    assert out["ARGS"] == "['--config', 'x.yaml']"
    assert out["VUNET"] == pkg + ".models.vunets" and out["NORMCONV"] == pkg + ".lib.modules"
    assert out["ACTNORM"] == "checkout" and out["FLOWLOSS"] == "checkout utils" and out["UTILS"] == "checkout utils"
    # the reference's other flows import these two by name and use them on 4-d inputs under autograd: they stay the checkout's
    assert out["FLOWBLOCKS"] == "checkout checkout"
    assert out["BEHAVIOR"] == f"{pkg}.models.flow.simple_flow {pkg}.models.pose_behavior_rnn checkout checkout"
    assert out["VGGLOSS"] == f"{pkg}.lib.losses {pkg}.models.imagenet_pretrained {pkg}.models.synth_discriminator"


def test_perceptual_vgg_adopts_a_torchvision_style_stack():
    """What the reference hands over is torchvision's vgg19: nn.Conv2d / nn.ReLU / nn.MaxPool2d at torchvision's indices."""
    import torch
    from torch import nn
    from behavior_driven_video_synthesis_amd.lib.modules import Conv2d
    from behavior_driven_video_synthesis_amd.models.imagenet_pretrained import VGG19_CFG, PerceptualVGG, _Marker
    layers, cin = [], 3
    for v in VGG19_CFG:
        if v == "M":
            layers.append(nn.MaxPool2d(2, 2))
        else:
            layers += [nn.Conv2d(cin, max(v // 16, 4), 3, padding=1), nn.ReLU(inplace=True)]
            cin = max(v // 16, 4)

    class TV(nn.Module):
        def __init__(self):
            super().__init__()
            self.features = nn.Sequential(*layers)
    tv = TV()
    pv = PerceptualVGG(tv, [1.0] * 6)
    mods = list(pv.vgg_layers)
    assert len(mods) == 37 and all(isinstance(m, (Conv2d, _Marker)) for m in mods)
    for i in (0, 2, 5, 28, 34):
        assert torch.equal(mods[i].weight, tv.features[i].weight) and torch.equal(mods[i].bias, tv.features[i].bias)
    assert [type(m).__name__ for m in mods[:5]] == ["Conv2d", "_Marker", "Conv2d", "_Marker", "_Marker"]
    assert not any(p.requires_grad for p in pv.vgg_layers.parameters())


def test_pretrained_model_flow_reads_a_reference_written_directory(tmp_path):
    """main.py:38-47 + experiments/experiment.py:39-95: ``--pretrained_model <dir>`` -- config.yaml beside ``reg_ckpt*.pth``.
    The directory under tests/golden/g8_pretrained was written by the REFERENCE's VunetAlter and torch.optim.Adam
    (make_golden.py g8_pretrained_dir); the loader must pick the newest checkpoint, restore strictly, carry
    Adam's state, the iteration and gamma, and mirror main.py's copy of config + checkpoints into the run directory."""
    import shutil
    import torch
    from conftest import GOLDEN, load_golden
    from behavior_driven_video_synthesis_amd.experiments.checkpoint import load_pretrained
    meta, _ = load_golden("g8_pretrained_outputs")
    src = tmp_path / "pretrained"
    shutil.copytree(os.path.join(GOLDEN, "g8_pretrained"), src)
    torch.save({"model": {"bogus": torch.zeros(1)}}, src / "reg_ckpt_model_1.pth")     # older: must not be chosen
    torch.save({"model": {"bogus": torch.zeros(1)}}, src / "discriminator_model_9.pth")  # other key: must not be chosen
    tr, cfg = load_pretrained(str(src), device="cpu", run_dir=str(tmp_path / "run"), vgg_synthetic=True, vgg_width_div=8,
                              total_steps=100)
    assert cfg["architecture"]["nf_max"] == 8 and tuple(cfg["training"]["adam_betas"]) == (0.5, 0.9)
    sd = tr.vunet.state_dict()
    assert len(sd) == meta["n_tensors"]
    for k, (s_, a_) in meta["checksums"].items():
        assert abs(float(sd[k].double().sum()) - s_) <= 1e-9 * max(a_, 1.0), k
    assert tr.iteration == meta["iteration"] and abs(float(tr.gamma) - meta["gamma"]) < 1e-12
    assert abs(tr.lr - cfg["training"]["lr"] * (1 - meta["iteration"] / 100)) < 1e-12        # schedule re-derived (:500-512)
    assert all(b.step == 2 and float(b.exp_avg.abs().sum()) > 0 for b in tr.optimizer.buckets)   # Adam's moments came along
    assert sorted(os.listdir(tmp_path / "run" / "ckpt")) == ["discriminator_model_9.pth", "reg_ckpt_model_1.pth",
                                                             "reg_ckpt_model_2.pth"]
    assert os.path.isfile(tmp_path / "run" / "config" / "config.yaml")
    with pytest.raises(FileNotFoundError):
        load_pretrained(str(tmp_path / "nowhere"), device="cpu")


def test_checkpoint_round_trip_keeps_the_regressor_file(tmp_path):
    """the reference keeps the regressor + its optimiser in a file of their own and restarts with
    ``_load_ckpt("regressor")`` (experiments/shape_and_pose_net.py:87-95, 486-497).  save_ckpt writes both files,
    ``load_ckpt(dir, "regressor")`` finds the second one, and ``restore`` brings a fresh trainer back to the saved state."""
    import copy
    import torch
    from behavior_driven_video_synthesis_amd.experiments.checkpoint import load_ckpt, restore, save_ckpt
    from behavior_driven_video_synthesis_amd.experiments.shape_and_pose_net import DEFAULT_CONFIG, ShapePoseNet
    cfg = copy.deepcopy(DEFAULT_CONFIG)
    cfg["data"]["spatial_size"] = 32
    cfg["architecture"].update(nf_start=4, nf_max=8)
    cfg["training"].update(train_regressor=True)
    a = ShapePoseNet(cfg, device="cpu", vgg_width_div=8, total_steps=50, vgg_synthetic=True)
    with torch.no_grad():   # pretend two optimiser steps happened (no kernels on this host)
        for tr_opt in (a.optimizer, a.optimizer_regressor):
            for b in tr_opt.buckets:
                b.step = 2
                b.exp_avg.normal_(generator=torch.Generator().manual_seed(b.numel))
                b.exp_avg_sq.uniform_(generator=torch.Generator().manual_seed(b.numel + 1))
        a.gamma.fill_(0.375)
    d = str(tmp_path / "ckpt")
    save_ckpt(d, "reg_ckpt", 2, a)
    assert sorted(os.listdir(d)) == ["reg_ckpt_checkpoint_2.pth", "regressor_checkpoint_2.pth"]
    reg_model, reg_opt = load_ckpt(d, "regressor")
    assert set(reg_model) == set(a.regressor.state_dict()) and reg_opt is not None and len(reg_opt["state"]) > 0
    torch.manual_seed(123)
    cfg2 = copy.deepcopy(cfg)
    cfg2["general"]["seed"] = 7        # a differently initialised trainer
    b_ = ShapePoseNet(cfg2, device="cpu", vgg_width_div=8, total_steps=50, vgg_synthetic=True)
    assert restore(d, b_)
    for (k, p), (_, q) in zip(a.regressor.state_dict().items(), b_.regressor.state_dict().items()):
        assert torch.equal(p, q), k
    for (k, p), (_, q) in zip(a.vunet.state_dict().items(), b_.vunet.state_dict().items()):
        assert torch.equal(p, q), k
    assert b_.iteration == 2 and float(b_.gamma) == 0.375
    for x, y in zip(a.optimizer_regressor.buckets, b_.optimizer_regressor.buckets):
        assert y.step == 2 and torch.equal(x.exp_avg, y.exp_avg) and torch.equal(x.exp_avg_sq, y.exp_avg_sq)
    assert not restore(str(tmp_path / "empty"), b_)


def test_bench_gpus_n_starts_ranks_or_fails_loudly():
    """bench.py --gpus N (the driver's call, no torchrun environment): with fewer than N GPUs visible it must exit
    non-zero without printing a result line -- never fall back to one rank (here: no GPU at all).  And started as one
    rank of a world whose size is not N it refuses as well."""
    import subprocess
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "VUNET_DP_FORCE")}
    need = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(max(2, need)), "--steps", "1",
                        "--warmup", "0"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr
    assert not any(l.startswith("{") for l in r.stdout.splitlines())
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "must agree" in r.stderr
    assert not any(l.startswith("{") for l in r.stdout.splitlines())


def _load_bench():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_watchdog_kills_a_silent_child_tree_and_keeps_a_talking_one(tmp_path):
    """bench.py --gpus N starts the ranks as a child tree and must not be able to wait forever (a replayed collective that
    never completes): a child that stops sending heartbeats is killed with everything it started; one that keeps talking and
    finishes is relayed.  Stub children stand in for the ranks."""
    import io
    import sys
    import time
    bench = _load_bench()
    hang = tmp_path / "hang.py"
    hang.write_text("import subprocess, sys, time\n"
                    "print('[bench heartbeat] settled', file=sys.stderr, flush=True)\n"
                    "subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(600)'])\n"   # a grandchild, as torchrun has
                    "time.sleep(600)\n")
    t0 = time.monotonic()
    rc, out, phase, reason = bench.run_with_watchdog([sys.executable, str(hang)], dict(os.environ), 20.0, 1.5, echo=io.StringIO())
    assert rc is None and phase == "settled" and "no heartbeat" in reason and "'settled'" in reason
    assert time.monotonic() - t0 < 15.0
    mute = tmp_path / "mute.py"
    mute.write_text("import time\ntime.sleep(600)\n")
    rc, out, phase, reason = bench.run_with_watchdog([sys.executable, str(mute)], dict(os.environ), 1.0, 30.0, echo=io.StringIO())
    assert rc is None and phase is None and "after the start" in reason
    ok = tmp_path / "ok.py"
    ok.write_text("import sys, time\n"
                  "for p in ('a', 'b', 'c'):\n"
                  "    print('[bench heartbeat] ' + p, file=sys.stderr, flush=True)\n"
                  "    time.sleep(0.6)\n"
                  "print('{\"metric\": 1}')\n")
    echo = io.StringIO()
    rc, out, phase, reason = bench.run_with_watchdog([sys.executable, str(ok)], dict(os.environ), 5.0, 1.0, echo=echo)
    assert rc == 0 and reason is None and phase == "c" and out.strip() == '{"metric": 1}' and "[bench heartbeat] b" in echo.getvalue()


def test_bench_graph_guard_prints_the_eager_line_when_the_replay_never_returns(tmp_path):
    """The in-process net of a multi-rank run: the eager measurement is taken first; if the graph phase does not come back the
    guard writes THAT line (with the reason) and leaves -- exit code 0, one JSON line."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "stuck.py"
    script.write_text(
        "import importlib.util, os, sys, time\n"
        f"spec = importlib.util.spec_from_file_location('b', {os.path.join(root, 'bench.py')!r})\n"
        "b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
        "g = b.GraphGuard(0.5, 1, {'metric': 'm', 'value': 700.0, 'config': {'hip_graph': True, 'fallback_reason': None}})\n"
        "time.sleep(600)\n")                          # "the replay": never returns
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0
    import json
    line = json.loads(r.stdout.strip())
    assert line["value"] == 700.0 and line["config"]["hip_graph"] is False and "did not complete" in line["config"]["fallback_reason"]
    ok = tmp_path / "fine.py"
    ok.write_text(script.read_text().replace("time.sleep(600)", "g.cancel(); time.sleep(1.0); print('{\"metric\": \"graph\"}')"))
    r = subprocess.run([sys.executable, str(ok)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == '{"metric": "graph"}'


def test_ctypes_descriptors_match_the_c_structs(tmp_path):
    """The host builds the kernels' descriptors with ctypes and uploads tables of them byte for byte: every mirrored struct of
    include/*.h must have the C compiler's size and field offsets (gcc here; the same LP64 layout hipcc uses)."""
    import ctypes
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from behavior_driven_video_synthesis_amd import seq, seq_train
    pairs = {"vunet_seq_linear_desc": seq.SeqLinearDesc, "vunet_seq_coupling_desc": seq.SeqCouplingDesc, "vunet_seq_lstm_desc": seq.SeqLstmDesc,
             "vunet_seq_adam_hp": seq_train.SeqAdamHp, "vunet_seq_dx_desc": seq_train.SeqDxDesc,
             "vunet_seq_coupling_bwd_desc": seq_train.SeqCouplingBwdDesc, "vunet_seq_dw_layer": seq_train.SeqDwLayer,
             "vunet_seq_actnorm_layer": seq_train.SeqActnormLayer, "vunet_seq_cell_bwd_desc": seq_train.SeqCellBwdDesc}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "vunet_hip.h"', '#include "vunet_seq_train.h"', '#include "vunet_seq_tiled.h"',
             "int main(void) {"]
    for cname, cls in pairs.items():
        lines.append(f'  printf("{cname} %zu", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf(" %zu", offsetof({cname}, {fname}));')
        lines.append('  printf("\\n");')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    for line in out.strip().splitlines():
        cname, size, *offs = line.split()
        cls = pairs[cname]
        assert ctypes.sizeof(cls) == int(size), (cname, ctypes.sizeof(cls), size)
        assert [getattr(cls, f).offset for f, _ in cls._fields_] == [int(o) for o in offs], cname


def test_input_gradient_slab_rule(monkeypatch):
    """``FlowTrainEngine._dx_split``: as many row ranges of W as keep a ``vunet_seq_dx`` launch within ONE round of workgroups on
    256 CUs, a slab at >= 64 rows, at most 16 -- and the LSTM's 17 column stripes on 15 slabs (255 workgroups), not 16 (272)."""
    from behavior_driven_video_synthesis_amd.seq_train import FlowTrainEngine as F
    monkeypatch.delenv("VUNET_SEQ_DX_TAIL", raising=False)
    want = {(2048, 2048, 2): 4, (2048, 512, 2): 16, (512, 2048, 2): 4, (4096, 1088, 1): 15, (4096, 1024, 1): 16, (1024, 1024, 2): 8,
            (64, 64, 2): 1, (192, 128, 2): 3}
    for (m, k, nets), s in want.items():
        assert F._dx_split(m, k, nets) == s, (m, k, nets)
    for m in (64, 128, 192, 512, 1024, 2048, 4096):
        for k in (64, 128, 512, 1088, 2048, 4096):
            for nets in (1, 2):
                s = F._dx_split(m, k, nets)
                wgs = (k // 64) * nets
                assert 1 <= s <= 16 and s <= m // 16 and (s == 1 or (wgs * s <= 256 and m // s >= 64)), (m, k, nets, s)
    monkeypatch.setenv("VUNET_SEQ_DX_TAIL", "1")
    assert F._dx_split(4096, 1088, 1) == 16 and F._dx_split(2048, 2048, 2) == 4
