"""Hot-path helpers of the reference's ``lib/utils.py`` (:33-48, :520-527, :658-668)."""
from __future__ import annotations

import numpy as np
from torch import nn


def n_parameters(model):
    """lib/utils.py:33-34."""
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def get_member(model, name):
    """lib/utils.py:37-43: attribute access through an optional nn.DataParallel wrapper."""
    module = model.module if isinstance(model, (nn.DataParallel, nn.parallel.DistributedDataParallel)) else model
    return getattr(module, name)


def toggle_grad(model, requires_grad):
    """lib/utils.py:46-48."""
    for p in model.parameters():
        p.requires_grad_(requires_grad)


def linear_var(act_it, start_it, end_it, start_val, end_val, clip_min, clip_max):
    """lib/utils.py:520-527."""
    act_val = float(end_val - start_val) / (end_it - start_it) * (act_it - start_it) + start_val
    return np.clip(act_val, a_min=clip_min, a_max=clip_max)


def scale_img(x):
    """lib/utils.py:658-668: [-1,1] -> [0,1], clamped (:666-667) -- VunetAlter's out_conv has no tanh."""
    import torch
    return torch.clamp((x + 1.0) / 2.0, 0.0, 1.0)


# ------------------------------------------------------------------------------------------------
# pose -> "stickman" image (lib/utils.py:325-512 make_joint_img) on the GPU rasteriser
# ------------------------------------------------------------------------------------------------
from collections import namedtuple  # noqa: E402

import torch  # noqa: E402

_JM_FIELDS = "body right_lines left_lines head_lines face rshoulder lshoulder headup"
JointModel = namedtuple("JointModel", _JM_FIELDS, defaults=(None, None, None))
"""The drawing fields of the reference's JointModel (lib/utils.py:23-26); any object with these attributes works."""

# data/human36m.py:137-148, keypoint_type keypoints_3d_world (the shipped Human3.6m config)
H36M_JOINT_MODEL = JointModel(body=[0, 14, 8, 11, 3],
                              right_lines=[(0, 1), (1, 2), (0, 14), (14, 15), (15, 16)],
                              left_lines=[(3, 4), (4, 5), (3, 11), (11, 12), (12, 13)],
                              head_lines=[(8, 9), (9, 10)], face=[])
# data/deepfashion.py:25-34 (18 OpenPose-style joints; no head lines: the neck branch draws shoulder-midpoint -> nose)
DEEPFASHION_JOINT_MODEL = JointModel(body=[8, 2, 5, 11], right_lines=[(10, 9), (9, 8), (2, 3), (3, 4)],
                                     left_lines=[(13, 12), (12, 11), (5, 6), (6, 7)], head_lines=[],
                                     face=[(0, 14), (0, 15), (14, 16), (15, 17)], rshoulder=2, lshoulder=5, headup=0)
# data/market.py:24-32 (BASELINE config 1)
MARKET_JOINT_MODEL = JointModel(body=[8, 9, 3, 2], right_lines=[(0, 1), (1, 2), (6, 7), (7, 8)],
                                left_lines=[(3, 4), (4, 5), (9, 10), (10, 11)], head_lines=[],
                                face=[(13, 14), (13, 15), (14, 16), (15, 17)], rshoulder=8, lshoulder=9, headup=13)

K_POLY, K_LINE, K_NECK, K_FACE, K_HEAD = 0, 1, 2, 3, 4   # command kinds of csrc/raster.hip


def _line_color(line_colors, group, nr):
    """(plane, colour) of make_joint_img's ``line_colors`` rule: the plane is the index of the non-zero entry of the
    line's colour triple, the colour that entry (lib/utils.py:365-373)."""
    triple = np.asarray(line_colors[group][nr])
    nz = np.nonzero(triple)[0]
    if len(nz) != 1:
        raise ValueError("line_colors entries must have exactly one non-zero channel (the reference calls int() on it)")
    return int(nz[0]), int(triple[nz[0]])


def stickman_draw_list(joint_model, line_colors=None, color_channel=None):
    """make_joint_img (lib/utils.py:325-512) as a draw list ``[kind, a, b, c, plane, colour]`` in the reference's draw
    order: body polygon (:345-355), right limbs (:357-380), left limbs (:382-405), then the neck line of a model
    without head lines (:407-433) or its head lines (:434-467), then the face lines (:468-505).  Defaults: polygon
    (0, 127, 255) on planes (0, 1, 2), right limbs 255 on plane 1, left limbs 255 on plane 0, head / neck / face
    lines 127 on planes 0 and 1; ``line_colors`` gives every limb / head / face line its own (plane, colour);
    ``color_channel`` draws everything with 255 on that one plane."""
    cmds = []
    one = color_channel is not None
    if len(joint_model.body) > 2:
        if one:
            cmds.append((K_POLY, 0, 0, 0, int(color_channel), 255))
        else:
            for plane, color in enumerate((0, 127, 255)):
                cmds.append((K_POLY, 0, 0, 0, plane, color))

    def lines(pairs, kind, group, default):
        for nr, (a, b) in enumerate(pairs):
            if one:
                cmds.append((kind, a, b, 0, int(color_channel), 255))
            elif line_colors is not None:
                plane, color = _line_color(line_colors, group, nr)
                cmds.append((kind, a, b, 0, plane, color))
            else:
                for plane, color in default:
                    cmds.append((kind, a, b, 0, plane, color))

    lines(joint_model.right_lines, K_LINE, 0, [(1, 255)])
    lines(joint_model.left_lines, K_LINE, 1, [(0, 255)])
    if len(joint_model.head_lines) == 0:
        if joint_model.rshoulder is None or joint_model.lshoulder is None or joint_model.headup is None:
            raise ValueError("a joint model without head lines needs rshoulder / lshoulder / headup (lib/utils.py:408-410)")
        rs, ls, hu = int(joint_model.rshoulder), int(joint_model.lshoulder), int(joint_model.headup)
        for plane, color in ([(int(color_channel), 255)] if one else [(0, 127), (1, 127)]):   # line_colors is not consulted here (:419-424)
            cmds.append((K_NECK, rs, ls, hu, plane, color))
    else:
        lines(joint_model.head_lines, K_HEAD, 2, [(0, 127), (1, 127)])
    lines(joint_model.face, K_FACE, 2, [(0, 127), (1, 127)])   # the reference indexes line_colors[2] for face lines too (:482)
    if len(cmds) > 32:
        raise ValueError("draw list longer than the rasteriser's 32 commands")
    return cmds


_raster_tables = {}


def make_joint_img_batch(img_shape, joints, joint_model=H36M_JOINT_MODEL, as_float=True, line_colors=None,
                         color_channel=None, thickness: int = 1):
    """Batched GPU make_joint_img: joints [B, J, 2] (x, y) device tensor -> [B, 3, H, W].

    ``as_float``: planes as fp32 in [-1, 1] (ToTensor then *2-1: data/base_dataset.py:183-190,
    data/__init__.py:23-24) -- the tensor VunetAlter takes as ``c``; else the raw uint8 planes.
    ``thickness``: cv2.line's thickness for every line of the frame (lib/utils.py:334-339)."""
    from .. import ops
    h, w = int(img_shape[0]), int(img_shape[1])
    joints = joints.to(torch.float32).contiguous()
    ops._dev(joints)
    lc_key = None if line_colors is None else repr(np.asarray(line_colors, dtype=object).tolist())
    key = (id(joint_model), joints.device, lc_key, color_channel)
    if key not in _raster_tables:
        cmds = torch.tensor(stickman_draw_list(joint_model, line_colors, color_channel), dtype=torch.int32,
                            device=joints.device).contiguous()
        body = torch.tensor(list(joint_model.body), dtype=torch.int32, device=joints.device)
        _raster_tables[key] = (body, cmds)
    body, cmds = _raster_tables[key]
    b, j = joints.shape[0], joints.shape[1]
    out = torch.empty(b, 3, h, w, device=joints.device, dtype=torch.float32 if as_float else torch.uint8)
    ops._call("vunet_stickman_raster_thick", ops._p(joints), b, j, ops._p(body), body.numel(), ops._p(cmds), cmds.shape[0],
              None if as_float else ops._p(out), ops._p(out) if as_float else None, h, w, int(thickness), ops._stream())
    return out


def make_joint_img(img_shape, joints, joint_model, line_colors=None, color_channel=None, scale_factor=None):
    """Reference signature (lib/utils.py:325-332): one frame, numpy in / HxWx3 uint8 numpy out (HxWx1 float mean when
    ``img_shape[-1] == 1``, :507-509).  ``scale_factor`` selects thickness ``img_shape[1] // scale_factor`` (:334-339, the
    ``stickman_scale`` of data/base_dataset.py:163-168): a thick cv2.line is OpenCV's ThickLine (FillConvexPoly quad +
    Circle caps), restated in csrc/raster.hip; cv2.line treats thickness 0 like 1."""
    thickness = int(img_shape[1] // scale_factor) if scale_factor is not None else 1
    if thickness < 0:
        raise ValueError("make_joint_img: negative line thickness")   # (cv2.line raises too)
    j = torch.as_tensor(np.asarray(joints, dtype=np.float32)).unsqueeze(0).cuda()
    out = make_joint_img_batch(img_shape[:2], j, joint_model, as_float=False, line_colors=line_colors,
                               color_channel=color_channel, thickness=max(thickness, 1))
    img = out[0].permute(1, 2, 0).cpu().numpy()
    if len(img_shape) > 2 and img_shape[-1] == 1:
        img = np.mean(img, axis=-1)[:, :, None]
    return img
