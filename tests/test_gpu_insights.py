"""-m gpu: VisualBackprop (reference insights/visual_backprop.py:10-53) through SheepLocalizer.predict(...,
return_visual_backprop=True), and the grayscale rois of SheepLocalizer(transform_rois_to_grayscale=True)
(sheep/sheep_localizer.py:65-68), against NumPy restatements built on the CPU oracle's activations."""
import numpy as np
import pytest
import torch

import loans_amd
from oracle import chainer_ops as C
from oracle import model as M
from tests.gpu_util import build_pair, dev, inputs, oracle_params

pytestmark = pytest.mark.gpu


def _ones_deconv(f, H, W, s, p):
    """F.deconvolution_2d(f, ones(1, 1, kh, kw), stride s, pad p, outsize (H, W)), kh = H + 2p - s (fh - 1)  (visual_backprop.py:29-37)"""
    B, fh, fw = f.shape
    kh, kw = H + 2 * p - s * (fh - 1), W + 2 * p - s * (fw - 1)
    full = np.zeros((B, s * (fh - 1) + kh, s * (fw - 1) + kw))
    for i in range(fh):
        for j in range(fw):
            full[:, s * i:s * i + kh, s * j:s * j + kw] += f[:, i:i + 1, j:j + 1]
    return full[:, p:p + H, p:p + W]


def _oracle_visual_backprop(lp, frames, crop):
    """the reference's walk on the oracle's activations (test mode): anchor = last feature map; nodes in reverse order:
    conv2, conv1 of every residual unit, the stem's max-pooling, conv1"""
    loc = M.Localizer(lp, crop, train=False)
    loc.forward(frames)
    nodes = [(C.prepare_images(frames), 7, 2, 3), (loc.stem_relu, 3, 2, 0)]                  # (input, k, s, p) in forward order
    h = C.max_pool_fwd(loc.stem_relu, 3, 2, 0)[0]
    for blk in loc.blocks:
        s1 = blk.c1.stride
        nodes.append((h, 3, s1, 1))
        nodes.append((blk.h1, 3, 1, 1))
        h = blk.out
    vis = loc.feat.mean(axis=1)
    for x, k, s, p in reversed(nodes):
        vis = _ones_deconv(vis, x.shape[2], x.shape[3], s, p) * x.mean(axis=1)
    lo, hi = vis.min(axis=(1, 2), keepdims=True), vis.max(axis=(1, 2), keepdims=True)
    return ((vis - lo) / (hi - lo))[:, None]


@pytest.mark.parametrize("hw", [(224, 224), (256, 232)])          # 256: res6 runs as well
def test_visual_backprop_against_oracle_walk(hw):
    crop = (32, 24)
    loc, _ = build_pair(91, crop)
    frames = inputs(92, 2, hw[0], hw[1], crop)[0]
    loc.finalize(torch.device('cuda', 0))
    # a fresh model's running statistics are (0, 1): in test mode nothing is normalised, activations grow to 1e2 per layer and
    # the product of the walk's 18 maps overflows fp32.  Thirty train-mode passes settle the statistics (decay 0.9) first.
    with loans_amd.using_config('enable_backprop', False):
        for _ in range(30):
            loc(dev(frames))
    lp = oracle_params(loc, np.float64)
    boxes, rois, scores, vis = loc.predict(list(frames), return_visual_backprop=True)
    assert vis.shape == (2, 1) + hw and vis.dtype == np.float32
    assert float(vis.min()) == 0.0 and abs(float(vis.max()) - 1.0) < 1e-6                  # min-max normalised per image
    ref = _oracle_visual_backprop(lp, frames.astype(np.float64), crop)
    # a product of 18 (20) positive-ish maps: compare after the normalisation, where rounding of the extremes enters once
    np.testing.assert_allclose(vis, ref, atol=2e-4)
    # the same call without the map leaves nothing behind
    assert loc.predict(list(frames))[3] is None and not hasattr(loc.visual_backprop_anchors[0], 'vbp_taps')


def test_visual_backprop_resnet50_runs():
    np.random.seed(93)
    loc = loans_amd.Resnet50SheepLocalizer((24, 24))
    frames = inputs(94, 2, 232, 232, (24, 24))[0]
    with loans_amd.using_config('enable_backprop', False):
        for _ in range(30):
            loc(dev(frames))
    vis = loc.predict(list(frames), return_visual_backprop=True)[3]
    assert vis.shape == (2, 1, 232, 232) and np.isfinite(vis).all() and vis.min() == 0.0
    # the anchor is res5 (sheep_localizer.py:156): 1 + 1 + 16 x 3 main-branch nodes were walked, not res6's
    assert len(loc.visual_backprop_anchors[0].vbp_taps) == 2 + 48


def test_grayscale_rois_forward_backward():
    crop = (16, 20)
    np.random.seed(95)
    loc = loans_amd.SheepLocalizer(crop, transform_rois_to_grayscale=True)
    loc.param_predictor.W.set_logical((2e-3 * np.random.RandomState(96).standard_normal((6, 512))).astype(np.float32))
    frames = inputs(97, 2, 64, 64, crop)[0]
    loc.finalize(torch.device('cuda', 0))
    lp = oracle_params(loc, np.float64)
    rois, points = loc(dev(frames))
    assert tuple(rois.shape) == (2, 1) + crop
    oloc = M.Localizer(lp, crop, train=True, rng=np.random.RandomState(0))
    o_rois, _ = oloc.forward(frames.astype(np.float64))
    gray = 0.299 * o_rois[:, 2:3] + 0.587 * o_rois[:, 1:2] + 0.114 * o_rois[:, 0:1]       # b, g, r = split_axis(rois, 3, 1)
    np.testing.assert_allclose(rois.data.cpu().numpy(), gray, atol=5e-4)
    # backward: d/d rois of sum(gray * g) reaches theta through the sampler
    g = np.random.RandomState(98).standard_normal(gray.shape)
    rois.grad = dev(g.astype(np.float32))
    loc.cleargrads()
    rois.backward()
    grads = {}
    g_rgb = np.concatenate([0.114 * g, 0.587 * g, 0.299 * g], axis=1)
    oloc.backward(g_rgb, None, grads)
    got, want = loc.param_predictor.b.grad_logical(), grads['param_predictor/b']
    np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-3 * np.abs(want).max())
