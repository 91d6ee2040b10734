"""Batched pose-sequence rendering (BASELINE config 5): 3-D poses -> camera projection -> stickman raster ->
``VunetAlter.transfer``, all on the GPU.

The reference renders a sequence frame by frame in Python (data/data_conversions_3d.py:1130-1185): per frame a
numpy projection (:588-605, :892-912), a CPU cv2 raster (lib/utils.py:325-512), an H2D copy, a batch-1
``synth_model.transfer`` and a D2H copy.  Here one raster launch draws every frame of the sequence and the
synthesis runs in batches; results are identical frame for frame (tests/test_hip_render.py).

``dtype="bf16"`` selects the precision BASELINE config 5 names for this loop; fp32 is the default.  The pose half of
``transfer`` (``du`` + ``dd``, the part that runs per frame) then executes on channel-blocked bf16 activations
(``render_blk.BlockedTransfer``, csrc/conv_blk.hip: bf16 operands and stored activations, fp32 accumulation);
``layout="nchw"`` keeps fp32 NCHW activations and only rounds the operands of the 3x3 convolutions
(``ops.inference_precision``, csrc/conv_bf16.hip) -- also what models outside ``BlockedTransfer.supported`` get.
``share_appearance=True`` encodes the appearance image once per sequence instead of once per frame (the
reference re-encodes the same image for every frame); with a fixed ``eps`` the frames are bit-identical either
way, without one the sequence shares a single posterior sample instead of drawing one per frame.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import ops
from .render_blk import engine_for
from .lib.utils import H36M_JOINT_MODEL, make_joint_img_batch, scale_img


def apply_affine_transform(x: torch.Tensor, m: torch.Tensor) -> torch.Tensor:
    """R x + t with M = [R, t] (3x4): data/data_conversions_3d.py:588-605.  x: [..., 3]."""
    ones = torch.ones_like(x[..., :1])
    return torch.cat([x, ones], dim=-1) @ m.transpose(-1, -2)


def camera_projection(poses: torch.Tensor, camera_parameters: Sequence[float]) -> torch.Tensor:
    """Pinhole projection with (f_x, x_0, f_y, y_0): data/data_conversions_3d.py:892-912.  poses: [..., J, 3]."""
    fx, x0, fy, y0 = [float(v) for v in camera_parameters]
    cam = torch.tensor([[fx, 0.0, x0], [0.0, fy, y0], [0.0, 0.0, 1.0]], dtype=poses.dtype, device=poses.device)
    p = poses / poses[..., -1:]
    return (p @ cam.T)[..., :-1]


def project_sequence(poses3d_world: torch.Tensor, extrinsics: torch.Tensor, intrinsics: Sequence[float],
                     image_size: Sequence[float], spatial_size: int) -> torch.Tensor:
    """[T, J, 3] world poses -> [T, J, 2] pixel keypoints at ``spatial_size`` (data/human36m.py:826-834)."""
    kps2d = camera_projection(apply_affine_transform(poses3d_world, extrinsics), intrinsics)
    scale = torch.tensor([float(spatial_size) / image_size[0], float(spatial_size) / image_size[1]],
                         dtype=kps2d.dtype, device=kps2d.device)
    return kps2d * scale


_side_streams = {}


def _side_stream(device):
    key = torch.device(device).index or 0
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


@torch.no_grad()
def render_sequence(vunet, app_img: torch.Tensor, kps2d: torch.Tensor, spatial_size: Optional[int] = None,
                    joint_model=H36M_JOINT_MODEL, chunk: int = 16, as_uint8: bool = True, dtype: str = "f32",
                    share_appearance: bool = False, eps=None, layout: str = "auto"):
    """Render T frames: ``app_img`` [1, C, H, W] (appearance), ``kps2d`` [T, J, 2] pixel keypoints.

    Returns (frames, stickmen): frames uint8 [T, H, W, 3] as the reference builds them
    (``scale_img(rgb) * 255``, channels last, :1160-1167; values are clamped to [0, 255] before the cast) or
    the raw fp32 [T, 3, H, W] output when ``as_uint8`` is False; stickmen fp32 [T, 3, H, W] in [-1, 1].
    """
    size = spatial_size or vunet.spatial_size
    was_training = vunet.training
    vunet.eval()
    if layout not in ("auto", "nchw", "blk"):
        raise ValueError(f"unknown activation layout {layout!r}")
    engine = engine_for(vunet) if (dtype == "bf16" and layout != "nchw") else None
    if layout == "blk" and engine is None:
        raise ValueError("layout='blk' needs dtype='bf16' and a model BlockedTransfer.supported covers")
    outs = []
    if engine is not None and share_appearance and app_img.is_cuda:
        # One appearance encoding per sequence: ~100 batch-1 launches that leave most of the chip idle.  They run on a
        # side stream, next to the raster and the pose pyramid of the first chunk, which need nothing from them.
        main, side = torch.cuda.current_stream(), _side_stream(app_img.device)
        side.wait_stream(main)
        with torch.cuda.stream(side), ops.inference_precision(dtype), ops.prepacked(vunet):
            code = engine.encode_code(vunet.appearance_code(app_img, eps))
        stick = make_joint_img_batch((size, size), kps2d.to(app_img.device), joint_model, as_float=True)
        for s in range(0, stick.shape[0], chunk):
            feats = engine.pose_features(stick[s:s + chunk])
            if s == 0:
                main.wait_stream(side)
                for t in code:
                    t.record_stream(main)
            outs.append(engine.decode(feats, code))
    else:
        stick = make_joint_img_batch((size, size), kps2d.to(app_img.device), joint_model, as_float=True)
        # the weights do not change inside a sequence: fold / pack every layer once (two launches) instead of per call
        with ops.inference_precision(dtype), ops.prepacked(vunet):
            code = vunet.appearance_code(app_img, eps) if share_appearance else None
            if code is not None and engine is not None:
                code = engine.encode_code(code)
            for s in range(0, stick.shape[0], chunk):
                c = stick[s:s + chunk]
                if code is None:
                    e = None if eps is None else [t.expand(c.shape[0], -1, -1, -1).contiguous() for t in eps]
                    frame_code = vunet.appearance_code(app_img.expand(c.shape[0], -1, -1, -1).contiguous(), e)
                    if engine is not None:
                        frame_code = engine.encode_code(frame_code)
                else:
                    frame_code = code
                outs.append(engine.transfer_code(frame_code, c) if engine is not None
                            else vunet.transfer_code(frame_code, c))
    rgb = torch.cat(outs, dim=0)
    vunet.train(was_training)
    if as_uint8:
        rgb = (scale_img(rgb) * 255.0).clamp_(0.0, 255.0).permute(0, 2, 3, 1).to(torch.uint8)
    return rgb, stick


# ------------------------------------------------------------------------------------------------
# BASELINE config 5 end to end: flow sample -> pose_behavior_rnn decode -> per-frame VUnet render
# ------------------------------------------------------------------------------------------------
class PoseCamera:
    """What turns decoded pose vectors into pixel keypoints: the dataset's normalisation statistics
    (``data_mean`` / ``data_std`` / ``dim_to_use``, data/data_conversions_3d.py:361-385), a camera's extrinsics [3, 4] and
    intrinsics (f_x, x_0, f_y, y_0), the camera's image size (h, w) and the synthesis resolution.  Held on the device."""

    def __init__(self, data_mean, data_std, dim_to_use, extrinsics, intrinsics, image_size, spatial_size, device="cuda"):
        import numpy as np
        mean, std = np.asarray(data_mean), np.asarray(data_std)
        self.f32_math = int(mean.dtype == np.float32 and std.dtype == np.float32)
        self.mean = torch.as_tensor(mean.astype(np.float64), device=device)
        self.std = torch.as_tensor(std.astype(np.float64), device=device)
        use = np.asarray(sorted(int(d) for d in dim_to_use), dtype=np.int32)
        self.dims = torch.as_tensor(use, device=device)
        self.n_use, self.dim = int(use.size), int(mean.size)
        ext = np.asarray(extrinsics, dtype=np.float64).reshape(3, 4)
        fx, x0, fy, y0 = [float(v) for v in intrinsics]
        # (joint_scaling = size / image size, applied to (x, y) with image_size given as (s[0], s[1]): :1139-1140)
        scale = [float(spatial_size) / float(image_size[0]), float(spatial_size) / float(image_size[1])]
        self.cam = torch.as_tensor(np.concatenate([ext.reshape(-1), [fx, x0, fy, y0], scale]), device=device)
        self.spatial_size = int(spatial_size)

    def project(self, poses: torch.Tensor) -> torch.Tensor:
        """[T, n_use] decoded (normalised) pose vectors -> [T, J, 2] pixel keypoints at the synthesis resolution."""
        ops._dev(poses)
        x = poses.reshape(-1, poses.shape[-1]).contiguous()
        if x.shape[1] != self.n_use:
            raise ValueError(f"pose vectors of {self.n_use} used dimensions expected, got {tuple(poses.shape)}")
        t, j = x.shape[0], self.dim // 3
        kps = torch.empty(t, j, 2, device=x.device, dtype=torch.float32)
        ops._call("vunet_seq_pose_project", ops._p(x), self.n_use, ops._p(self.dims), ops._p(self.mean), ops._p(self.std), self.dim,
                  self.f32_math, ops._p(self.cam), ops._p(kps), t, j, ops._stream())
        return kps


@torch.no_grad()
def behavior_video(flow, net, vunet, app_img: torch.Tensor, start_poses: torch.Tensor, length: int, camera: PoseCamera,
                   z: Optional[torch.Tensor] = None, start_frame: int = -1, joint_model=H36M_JOINT_MODEL, dtype: str = "bf16",
                   share_appearance: bool = True, chunk: int = 16):
    """One synthesised sequence per row of ``start_poses`` [B, T, n_kps] (experiments/behavior_net.py:1173-1184 followed by
    data/data_conversions_3d.py:1130-1185): behaviour codes b = flow.reverse(z) with z ~ N(0, 1) (or the given ``z`` [B, C]),
    the decoder's roll-out of ``length`` poses from ``start_poses[:, start_frame]``, camera projection, stickman raster and
    ``VunetAlter.transfer`` of ``app_img`` [1, 3, H, W] for every frame.

    Returns (frames uint8 [B, length, H, W, 3], poses [B, length, n_kps], keypoints [B, length, J, 2])."""
    bsz = start_poses.shape[0]
    if z is None:
        z = torch.randn(bsz, flow.in_channels, device=start_poses.device)
    b = flow.reverse(z)
    b = b.reshape(bsz, -1)
    poses, *_ = net.generate_seq(b, start_poses, len=length, start_frame=start_frame % start_poses.shape[1])
    kps = camera.project(poses.reshape(bsz * length, -1)).reshape(bsz, length, -1, 2)
    frames = []
    for i in range(bsz):
        rgb, _ = render_sequence(vunet, app_img, kps[i], spatial_size=camera.spatial_size, joint_model=joint_model, chunk=chunk,
                                 dtype=dtype, share_appearance=share_appearance)
        frames.append(rgb)
    return torch.stack(frames), poses, kps
