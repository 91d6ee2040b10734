"""-m gpu: batched sequence rendering == the reference's frame-by-frame loop (raster -> batch-1 transfer)."""
import numpy as np
import pytest
import torch

from hip_parity_utils import assert_close
from synth import synth_image, synth_state_dict

pytestmark = pytest.mark.gpu


def test_render_sequence_matches_per_frame_loop_and_oracle_raster():
    from oracle import stickman as S
    from oracle import vunet_oracle as O
    from behavior_driven_video_synthesis_amd.lib.utils import H36M_JOINT_MODEL, stickman_draw_list
    from behavior_driven_video_synthesis_amd.models.vunets import VunetAlter
    from behavior_driven_video_synthesis_amd.render import project_sequence, render_sequence
    cfg = dict(spatial_size=64, bottleneck_factor=2, box_factor=2, n_scales=0, n_latent_scales=2,
               conv_layer_type="l1", nf_start=8, nf_max=16, subpixel_upsampling=True, dropout_prob=0.05)
    net = VunetAlter(**cfg)
    sd = synth_state_dict({k: list(v.shape) for k, v in net.state_dict().items()}, 4)
    net.load_state_dict(sd)
    net = net.cuda()
    rng = np.random.default_rng(0)
    T = 7
    world = torch.from_numpy(rng.normal(0, 0.4, size=(T, 17, 3))).float() + torch.tensor([0.0, 0.0, 4.0])
    extr = torch.tensor([[1.0, 0, 0, 0.1], [0, 1.0, 0, -0.05], [0, 0, 1.0, 0.3]])
    kps = project_sequence(world, extr, (1100.0, 500.0, 1100.0, 500.0), (1000, 1000), 64)
    app = synth_image("app", (1, 3, 64, 64), 4).cuda()
    eps = None
    torch.manual_seed(0)
    frames, stick = render_sequence(net, app, kps.cuda(), chunk=3, as_uint8=False)
    # stickmen: bit-exact vs the CPU raster oracle
    want = S.raster(kps.numpy(), H36M_JOINT_MODEL.body, stickman_draw_list(H36M_JOINT_MODEL), 64, 64)
    want_f = (want.astype(np.float32) / np.float32(255.0)) * np.float32(2.0) - np.float32(1.0)
    assert np.array_equal(stick.cpu().numpy(), want_f)
    # transfer uses the posterior means (no sampling in the output path): per-frame oracle loop
    for t in range(T):
        ref = O.vunet_alter_transfer(sd, cfg, app.cpu(), torch.from_numpy(want_f[t:t + 1]))
        assert_close(frames[t:t + 1], ref, name=f"frame {t}")
    u8, _ = render_sequence(net, app, kps.cuda(), chunk=4, as_uint8=True)
    assert u8.shape == (T, 64, 64, 3) and u8.dtype == torch.uint8


def test_render_bf16_precision_and_shared_appearance():
    """dtype="bf16" (BASELINE config 5): bf16 operands / fp32 accumulate on the 3x3 layers, PSNR vs the fp32 render
    well above the 0.1 dB budget; share_appearance=True (one appearance encoding per sequence) renders the same frames."""
    from hip_parity_utils import psnr
    from behavior_driven_video_synthesis_amd import ops
    from behavior_driven_video_synthesis_amd.models.vunets import VunetAlter
    from behavior_driven_video_synthesis_amd.render import render_sequence
    cfg = dict(spatial_size=64, bottleneck_factor=2, box_factor=2, n_scales=0, n_latent_scales=2,
               conv_layer_type="l1", nf_start=16, nf_max=32, subpixel_upsampling=True, dropout_prob=0.05)
    net = VunetAlter(**cfg)
    net.load_state_dict(synth_state_dict({k: list(v.shape) for k, v in net.state_dict().items()}, 9))
    net = net.cuda().eval()
    app = synth_image("app16", (1, 3, 64, 64), 9).cuda()
    rng = np.random.default_rng(3)
    kps = torch.from_numpy(rng.uniform(6, 58, size=(5, 17, 2))).float().cuda()
    with torch.no_grad():
        shapes = [tuple(m.shape) for m in net.appearance_code(app)]
    eps = [synth_image(f"eps{i}", s, 9).cuda() for i, s in enumerate(shapes)]

    f32, _ = render_sequence(net, app, kps, chunk=2, as_uint8=False, eps=eps)
    shared, _ = render_sequence(net, app, kps, chunk=3, as_uint8=False, eps=eps, share_appearance=True)
    assert_close(shared, f32, rtol=1e-4, atol=1e-5, name="shared appearance code")

    ops.profile_start()
    bf16, _ = render_sequence(net, app, kps, chunk=5, as_uint8=False, eps=eps, dtype="bf16", layout="nchw")
    fam = ops.profile_stop()
    assert "conv_bf16_fwd" in fam and fam["conv_bf16_fwd"]["n"] >= 8        # the bf16 kernel really ran
    scale = float(f32.abs().max())
    db = psnr(bf16, f32, peak=2 * scale)
    assert db >= 40.0, db
    assert float((bf16 - f32).abs().max()) <= 0.03 * scale
    # the switch is scoped: outside the context the same call is fp32 again, bit for bit
    again, _ = render_sequence(net, app, kps, chunk=2, as_uint8=False, eps=eps)
    assert torch.equal(again, f32)
