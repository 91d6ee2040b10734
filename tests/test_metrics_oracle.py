"""CPU: the SSIM oracle (oracle/metrics_oracle.py, a restatement of skimage's structural_similarity as the reference
calls it, lib/metrics.py:94-107) against closed-form cases -- scikit-image itself is not in the image."""
import numpy as np

from oracle import metrics_oracle as M


def test_window_is_the_11_tap_sigma_1p5_gaussian():
    w = M.gaussian_window()
    assert w.shape == (11,) and abs(w.sum() - 1.0) < 1e-15 and np.allclose(w, w[::-1])
    assert abs(w[5] / w[4] - np.exp(0.5 / 2.25)) < 1e-12


def test_identical_images_score_one_and_constants_follow_the_formula():
    rng = np.random.default_rng(0)
    x = rng.random((3, 24, 40))
    assert abs(M.ssim_image(x, x) - 1.0) < 1e-12
    # two constant planes a, b: all variances vanish, S = (2ab + C1) / (a^2 + b^2 + C1) everywhere
    a, b = 0.3, 0.7
    want = (2 * a * b + 1e-4) / (a * a + b * b + 1e-4)
    got = M.ssim_plane(np.full((20, 20), a), np.full((20, 20), b))
    assert abs(got - want) < 1e-12


def test_symmetry_range_and_crop():
    rng = np.random.default_rng(1)
    x, y = rng.random((16, 30)), rng.random((16, 30))
    assert abs(M.ssim_plane(x, y) - M.ssim_plane(y, x)) < 1e-14
    assert -1.0 <= M.ssim_plane(x, y) <= 1.0
    # only the 5-pixel-cropped interior counts: changing the outermost ring far from the interior's 11x11 windows ...
    big_x, big_y = rng.random((40, 40)), rng.random((40, 40))
    s0 = M.ssim_plane(big_x, big_y)
    # ... is impossible (every border pixel is inside some interior window), but data_range rescaling is exact:
    assert abs(M.ssim_plane(255 * big_x, 255 * big_y, data_range=255.0) - s0) < 1e-12
    assert M.psnr(big_x, big_x) == float("inf") and abs(M.psnr(np.zeros(4), np.full(4, 0.1)) - 20.0) < 1e-9


# ---- FID / Inception score: the oracle AND the product's host-side statistics against the reference's own functions
def _g7_inputs():
    import os, sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from conftest import load_golden
    from synth import seeded_randn, synth_image
    meta, _ = load_golden("g7_metrics")
    seed = meta["seed"]
    a = seeded_randn("fid.a", (400, 24), seed).double().numpy() @ seeded_randn("fid.ma", (24, 24), seed).double().numpy()
    b = seeded_randn("fid.b", (300, 24), seed).double().numpy() @ seeded_randn("fid.mb", (24, 24), seed).double().numpy() + 0.3
    d = seeded_randn("fid.d", (10, 24), seed).double().numpy()
    w = seeded_randn("is.w", (3 * 8 * 8, 1000), seed) * 3.0
    imgs = synth_image("is.imgs", (96, 3, 32, 32), seed)
    logits = torch.nn.functional.adaptive_avg_pool2d(imgs, 8).flatten(1) @ w
    return meta, a, b, d, imgs, w, logits


def test_frechet_distance_vs_reference_golden():
    from oracle import metrics_oracle as M
    from behavior_driven_video_synthesis_amd.lib import metrics as P
    meta, a, b, d, *_ = _g7_inputs()
    for fn in (M.fid_from_features, P.fid_from_features):
        assert abs(fn(a, b) - meta["fid"]) <= 1e-6 * meta["fid"]
        assert abs(fn(a, a)) <= 1e-6
        assert abs(fn(a, d) - meta["fid_singular"]) <= 1e-5 * meta["fid_singular"]   # rank-deficient covariance


def test_inception_score_vs_reference_golden():
    import torch
    from oracle import metrics_oracle as M
    from behavior_driven_video_synthesis_amd.lib import metrics as P
    meta, a, b, d, imgs, w, logits = _g7_inputs()
    probs = torch.softmax(logits.double(), dim=1).numpy()
    for splits in (1, 4):
        want = meta[f"is_{splits}"]
        for got in (M.inception_score_from_probs(probs, splits), P.inception_score_from_probs(probs, splits)):
            assert abs(got[0] - want[0]) <= 1e-5 * want[0] and abs(got[1] - want[1]) <= 1e-5 * max(want[0], 1.0)
        # the driver with a pluggable classifier (what the user supplies in place of torchvision's inception_v3)
        got = P.inception_score(imgs, lambda x: torch.nn.functional.adaptive_avg_pool2d(x, 8).flatten(1) @ w,
                                batch_size=16, splits=splits)
        assert abs(got[0] - want[0]) <= 1e-5 * want[0] and abs(got[1] - want[1]) <= 1e-5 * max(want[0], 1.0)


def test_scale_img_clamps_like_the_reference():
    import torch
    from oracle import metrics_oracle as M
    from behavior_driven_video_synthesis_amd.lib.utils import scale_img
    x = torch.tensor([-1.7, -1.0, 0.0, 0.5, 1.0, 2.3])            # VunetAlter's out_conv has no tanh: values leave [-1, 1]
    want = [0.0, 0.0, 0.5, 0.75, 1.0, 1.0]
    assert scale_img(x).tolist() == want and M.scale_img(x.numpy()).tolist() == want
