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

JointModel = namedtuple("JointModel", "body right_lines left_lines head_lines face")

# data/human36m.py:137-148, keypoint_type keypoints_3d_world (the shipped Human3.6m config)
H36M_JOINT_MODEL = JointModel(body=[0, 14, 8, 11, 3],
                              right_lines=[(0, 1), (1, 2), (0, 14), (14, 15), (15, 16)],
                              left_lines=[(3, 4), (4, 5), (3, 11), (11, 12), (12, 13)],
                              head_lines=[(8, 9), (9, 10)], face=[])


def stickman_draw_list(joint_model):
    """Draw list of make_joint_img's default branch (no line_colors / color_channel, thickness 1), in the
    reference's draw order: body polygon (0,127,255) on planes (0,1,2) (:345-355), right limbs 255 on plane 1
    (:357-380), left limbs 255 on plane 0 (:382-405), head lines 127 on planes 0 and 1 (:434-462)."""
    if len(joint_model.head_lines) == 0 or len(joint_model.face) > 0:
        raise NotImplementedError("neck / face branches of make_joint_img (lib/utils.py:407-433,468-505) are not built")
    cmds = []
    if len(joint_model.body) > 2:
        for plane, color in enumerate((0, 127, 255)):
            cmds.append((0, 0, 0, plane, color))
    for a, b in joint_model.right_lines:
        cmds.append((1, a, b, 1, 255))
    for a, b in joint_model.left_lines:
        cmds.append((1, a, b, 0, 255))
    for a, b in joint_model.head_lines:
        cmds.append((1, a, b, 0, 127))
        cmds.append((1, a, b, 1, 127))
    return cmds


_raster_tables = {}


def make_joint_img_batch(img_shape, joints, joint_model=H36M_JOINT_MODEL, as_float=True):
    """Batched GPU make_joint_img: joints [B, J, 2] (x, y) device tensor -> [B, 3, H, W].

    ``as_float``: planes as fp32 in [-1, 1] (ToTensor then *2-1: data/base_dataset.py:183-190,
    data/__init__.py:23-24) -- the tensor VunetAlter takes as ``c``; else the raw uint8 planes."""
    from .. import ops
    h, w = int(img_shape[0]), int(img_shape[1])
    joints = joints.to(torch.float32).contiguous()
    ops._dev(joints)
    key = (id(joint_model), joints.device)
    if key not in _raster_tables:
        cmds = torch.tensor(stickman_draw_list(joint_model), dtype=torch.int32, device=joints.device).contiguous()
        body = torch.tensor(list(joint_model.body), dtype=torch.int32, device=joints.device)
        _raster_tables[key] = (body, cmds)
    body, cmds = _raster_tables[key]
    b, j = joints.shape[0], joints.shape[1]
    out = torch.empty(b, 3, h, w, device=joints.device, dtype=torch.float32 if as_float else torch.uint8)
    ops._call("vunet_stickman_raster", ops._p(joints), b, j, ops._p(body), body.numel(), ops._p(cmds), cmds.shape[0],
              None if as_float else ops._p(out), ops._p(out) if as_float else None, h, w, ops._stream())
    return out


def make_joint_img(img_shape, joints, joint_model, line_colors=None, color_channel=None, scale_factor=None):
    """Reference signature (lib/utils.py:325-332): one frame, numpy in / HxWx3 uint8 numpy out."""
    if line_colors is not None or color_channel is not None or scale_factor is not None:
        raise NotImplementedError("only the default branch (thickness 1, fixed colours) is built")
    j = torch.as_tensor(np.asarray(joints, dtype=np.float32)).unsqueeze(0).cuda()
    out = make_joint_img_batch(img_shape[:2], j, joint_model, as_float=False)
    return out[0].permute(1, 2, 0).cpu().numpy()
