"""Known-answer tests for the CPU oracle, derived from reference *source* alone
(SURVEY §8c list; the reference has no tests of its own, so these are the only
reference-anchored pins)."""
import numpy as np
import pytest

from oracle import chainer_ops as C
from oracle import model as M
from loans_amd.datasets import synthetic


def _fresh(seed=0, w_std=0.0):
    rng = np.random.RandomState(seed)
    return M.init_localizer_params(rng, predictor_w_std=w_std)


def test_kat1_fresh_localizer_theta_points_bbox():
    # sheep_localizer.py:28-33 : W = 0, b = [.8,0,0,0,.8,0]  =>  theta fixed for any input
    p = _fresh()
    imgs = synthetic.make_frames(3, 2, 64, 64)
    loc = M.Localizer(p, (16, 16), train=True, rng=np.random.RandomState(0))
    rois, points = loc.forward(imgs)
    assert rois.shape == (2, 3, 16, 16) and points.shape == (2, 2, 16, 16)
    np.testing.assert_allclose(loc.theta, np.tile(np.array([[.8, 0, 0], [0, .8, 0]], np.float32), (2, 1, 1)), atol=1e-6)
    lin = np.linspace(-1, 1, 16, dtype=np.float32)
    np.testing.assert_allclose(points[:, 0], np.broadcast_to(0.8 * lin[None, None, :], (2, 16, 16)), atol=1e-6)
    np.testing.assert_allclose(points[:, 1], np.broadcast_to(0.8 * lin[None, :, None], (2, 16, 16)), atol=1e-6)
    # sheep_localizer.py:84-97 on a 224x224 frame -> [22.4, 22.4, 201.6, 201.6]
    bb = loc.corners_px(points, (224, 224))
    np.testing.assert_allclose(bb, np.tile(np.array([22.4, 22.4, 201.6, 201.6], np.float32), (2, 1)), rtol=1e-5)


def test_kat2_regularisers():
    th = tw = 8
    theta = np.tile(np.array([[.8, 0, 0], [0, .8, 0]], np.float32), (3, 1, 1))
    grid, _ = C.st_grid_fwd(theta, (th, tw))
    assert C.direction_loss(grid, (224, 224))[0] == 0
    assert C.out_of_image_loss(grid)[0] == 0
    theta = np.tile(np.array([[1.5, 0, 0], [0, 1.0, 0]], np.float32), (3, 1, 1))
    grid, _ = C.st_grid_fwd(theta, (th, tw))
    # TL_x = -1.5, TR_x = 1.5 -> 0.5 + 0.5 per image; it is a SUM over the batch (utils.py:315)
    np.testing.assert_allclose(C.out_of_image_loss(grid)[0], 3.0, rtol=1e-6)
    # mirrored grid: TL_x > TR_x and TL_y > BL_y by 1.6 * size / 2 each
    theta = np.tile(np.array([[-.8, 0, 0], [0, -.8, 0]], np.float32), (3, 1, 1))
    grid, _ = C.st_grid_fwd(theta, (th, tw))
    np.testing.assert_allclose(C.direction_loss(grid, (100, 200))[0], 0.8 * 100 + 0.8 * 200, rtol=1e-5)


@pytest.mark.parametrize("train", [True, False])
def test_kat3_rotation_dropout_zeroes_rotation(train):
    theta = np.random.RandomState(1).randn(4, 2, 3).astype(np.float32)
    mask = C.rotation_dropout_mask(theta, 0.0, train, np.random.RandomState(0))
    out = theta * mask
    assert np.all(out[:, 0, 1] == 0) and np.all(out[:, 1, 0] == 0)
    keep = np.ones((2, 3), bool); keep[0, 1] = keep[1, 0] = False
    assert np.array_equal(out[:, keep], theta[:, keep])


def test_kat4_fresh_model_backbone_grads_zero():
    # param_predictor.W == 0 -> no gradient reaches the backbone on step 1
    p = _fresh()
    dp = M.init_assessor_params(np.random.RandomState(1), (16, 16))
    frames = synthetic.make_frames(5, 2, 64, 64)
    real, labels = synthetic.make_assessor_batch(6, 2, 16, 16, src=64)
    out = M.update_core(p, dp, M.AdamAMSGrad(p), M.AdamAMSGrad(dp), frames, real, labels, (16, 16),
                        rng=np.random.RandomState(0), return_grads=True)
    for k, g in out['loc_grads'].items():
        if k.startswith('param_predictor'):
            continue
        assert not np.any(g), k
    assert np.any(out['loc_grads']['param_predictor/b'])


def test_kat5_sampler_identity_and_fade():
    rng = np.random.RandomState(0)
    x = rng.rand(2, 3, 12, 10).astype(np.float32)
    theta = np.tile(np.array([[1, 0, 0], [0, 1, 0]], np.float32), (2, 1, 1))
    grid, _ = C.st_grid_fwd(theta, (12, 10))
    np.testing.assert_allclose(C.st_sampler_fwd(x, grid), x, atol=2e-6)
    # scale 2: samples beyond one pixel outside the image are exactly zero
    theta = np.tile(np.array([[2, 0, 0], [0, 2, 0]], np.float32), (2, 1, 1))
    grid, _ = C.st_grid_fwd(theta, (12, 10))
    y = C.st_sampler_fwd(x, grid)
    u = (grid[:, 0] + 1) * (10 - 1) / 2
    v = (grid[:, 1] + 1) * (12 - 1) / 2
    far = (u < -1) | (u > 10) | (v < -1) | (v > 12)
    assert far.any()
    assert np.all(y[np.broadcast_to(far[:, None], y.shape)] == 0)


def test_kat6_shapes():
    assert C.conv_outsize(224, 7, 2, 3) == 112
    assert C.conv_outsize(112, 3, 2, 0, cover_all=True) == 56       # not 55
    assert C.conv_outsize(512, 7, 2, 3) == 256 and C.conv_outsize(256, 3, 2, 0, cover_all=True) == 128
    a = C.conv_outsize(75, 4, 2, 1); b = C.conv_outsize(a, 4, 2, 1)
    assert (a, b) == (37, 18)
    assert M.init_assessor_params(np.random.RandomState(0))['l4/W'].shape == (1, 41472)
    y, idx = C.max_pool_fwd(np.zeros((1, 2, 112, 112), np.float32))
    assert y.shape == (1, 2, 56, 56)


def test_kat7_preprocess_truncation():
    k = np.arange(256, dtype=np.float32)
    x = np.zeros((1, 3, 16, 16), np.float32)
    x[0, 0].flat[:] = (k / np.float32(255))            # R = k/255
    x[0, 1].flat[:] = ((k + 0.5) / np.float32(255)).clip(0, 1)  # G: truncation, not rounding
    out = C.prepare_images(x)
    # BGR flip: channel 2 of the output is R, channel 1 is G
    np.testing.assert_array_equal(out[0, 2].ravel(), k - np.float32(123.152))
    g_expected = np.minimum(k, 255) - np.float32(115.903)
    g_expected[255] = 255 - np.float32(115.903)
    np.testing.assert_array_equal(out[0, 1].ravel(), g_expected)
    np.testing.assert_array_equal(out[0, 0].ravel(), np.full(256, -np.float32(103.063)))


def test_kat8_assessor_freeze():
    p = _fresh(w_std=1e-3)
    dp = M.init_assessor_params(np.random.RandomState(1), (16, 16))
    before = {k: v.copy() for k, v in dp.items()}
    frames = synthetic.make_frames(5, 2, 64, 64)
    real, labels = synthetic.make_assessor_batch(6, 2, 16, 16, src=64)
    M.update_core(p, dp, M.AdamAMSGrad(p), M.AdamAMSGrad(dp), frames, real, labels, (16, 16),
                  freeze_discriminator=True, rng=np.random.RandomState(0))
    for k in dp:
        assert np.array_equal(dp[k], before[k]), k


def test_adam_first_step_is_chainer_placement():
    # step 1: m = .1 g, v = .001 g^2, lr_t = a*sqrt(.001)/.1 ; eps OUTSIDE the bias correction
    g = np.array([1e-3, -2.0, 1e-9], np.float64)
    p = np.zeros(3); m = np.zeros(3); v = np.zeros(3); vh = np.zeros(3)
    C.adam_amsgrad_update(p, g, m, v, vh, 1, alpha=1e-3)
    lr = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    np.testing.assert_allclose(p, -lr * 0.1 * g / (np.sqrt(0.001 * g * g) + 1e-8), rtol=1e-10)
    # amsgrad: vhat never decreases
    C.adam_amsgrad_update(p, np.zeros(3), m, v, vh, 2, alpha=1e-3)
    assert np.all(vh >= v)


def test_kat9_ties_follow_chainers_maximum_and_absolute():
    """Chainer 4.1 routes the gradient of F.maximum(x1, x2) with `x1 >= x2` (F.minimum: `x1 <= x2`): at an exact tie it goes to
    the first argument.  The regularisers call them as F.maximum(distance, zeros) / F.maximum(bottom_loss, zeros)
    (common/utils.py:169,175,313), so a corner sitting EXACTLY on the image border, or a box of exactly zero extent, still
    receives the gradient; F.absolute(F.minimum(top_loss, zeros)) (:312) does not: Absolute's backward is sign(x) gy and
    sign(0) = 0.  theta = identity puts TR_x and BL_y exactly on +1 and TL_x, TL_y exactly on -1."""
    th, tw, B = 5, 6, 3
    theta = np.tile(np.array([[1, 0, 0], [0, 1, 0]], np.float32), (B, 1, 1))
    grid, _ = C.st_grid_fwd(theta, (th, tw))
    assert grid[0, 0, 0, tw - 1] == 1.0 and grid[0, 1, th - 1, 0] == 1.0 and grid[0, 0, 0, 0] == -1.0 and grid[0, 1, 0, 0] == -1.0
    loss, gg = C.out_of_image_loss(grid)
    assert loss == 0
    want = np.zeros_like(grid)
    want[:, 0, 0, tw - 1] = 1.0          # TR_x - 1 == 0: max(bottom_loss, 0) passes the gradient at the tie
    want[:, 1, th - 1, 0] = 1.0          # BL_y - 1 == 0
    np.testing.assert_array_equal(gg, want)                   # TL_x + 1 == 0, TL_y + 1 == 0: sign(0) = 0, nothing
    # a box of zero height: TL_y - BL_y == 0 exactly -> the direction term passes H / (2 B) to TL_y and its negative to BL_y
    theta = np.tile(np.array([[1, 0, 0], [0, 0, 0.25]], np.float32), (B, 1, 1))
    grid, _ = C.st_grid_fwd(theta, (th, tw))
    loss, gg = C.direction_loss(grid, (48, 64))
    assert loss == 0
    want = np.zeros_like(grid)
    want[:, 1, 0, 0] = 48 / 2.0 / B
    want[:, 1, th - 1, 0] = -48 / 2.0 / B
    np.testing.assert_array_equal(gg, want)                   # (TL_x - TR_x = -64 < 0: no horizontal term)
