"""-m gpu: the GPU stickman rasteriser (closed-form per pixel) is bit-exact against the sequential CPU restatement."""
import numpy as np
import pytest
import torch

from oracle import stickman as S

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("size,spread,seed", [(256, 40, 0), (64, 30, 1), (128, 90, 2), (32, 6, 3)])
def test_raster_bit_exact_vs_oracle(size, spread, seed):
    from behavior_driven_video_synthesis_amd.lib.utils import H36M_JOINT_MODEL, make_joint_img_batch, stickman_draw_list
    rng = np.random.default_rng(seed)
    b = 24
    kps = rng.normal(size / 2, spread, size=(b, 17, 2)).astype(np.float32)   # some joints negative / outside the frame
    kps[3, 5] = (-3.0, 10.0)                                                  # invalid joint -> its lines vanish
    kps[4, [0, 14, 8]] = -1.0                                                 # polygon left with 2 valid points
    want = S.raster(kps, H36M_JOINT_MODEL.body, stickman_draw_list(H36M_JOINT_MODEL), size, size)
    got = make_joint_img_batch((size, size), torch.from_numpy(kps).cuda(), H36M_JOINT_MODEL, as_float=False)
    assert np.array_equal(got.cpu().numpy(), want)
    assert want.any()
    f = make_joint_img_batch((size, size), torch.from_numpy(kps).cuda(), H36M_JOINT_MODEL, as_float=True).cpu().numpy()
    assert np.array_equal(f, (want.astype(np.float32) / np.float32(255.0)) * np.float32(2.0) - np.float32(1.0))


def test_make_joint_img_reference_signature():
    from behavior_driven_video_synthesis_amd.lib.utils import H36M_JOINT_MODEL, make_joint_img, stickman_draw_list
    rng = np.random.default_rng(5)
    kps = rng.normal(64, 25, size=(17, 2)).astype(np.float32)
    img = make_joint_img([128, 128, 3], kps, H36M_JOINT_MODEL)
    assert img.shape == (128, 128, 3) and img.dtype == np.uint8
    want = S.raster(kps[None], H36M_JOINT_MODEL.body, stickman_draw_list(H36M_JOINT_MODEL), 128, 128)[0]
    assert np.array_equal(img, want.transpose(1, 2, 0))
