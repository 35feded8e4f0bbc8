"""Oracle for the ResNet-50 localizer against the independent torch-CPU composition (fp64)."""
import numpy as np
import pytest
import torch

from oracle import chainer_ops as C
from oracle import model as M
from loans_amd.datasets import synthetic
from tests import torch_reference as T


@pytest.mark.parametrize("hw", [(64, 64), (232, 226)])
def test_resnet50_localizer_forward_and_grads_match_torch(hw):
    crop = (12, 10)
    rng = np.random.RandomState(5)
    lp = M.cast_params(M.init_resnet50_localizer_params(rng, predictor_w_std=2e-2), np.float64)
    for k in lp:
        if k.endswith('/gamma'):
            lp[k] = 1 + 0.1 * rng.standard_normal(lp[k].shape)
        if k.endswith('/beta'):
            lp[k] = 0.1 * rng.standard_normal(lp[k].shape)
    B = 3 if hw[0] <= 64 else 2
    frames = synthetic.make_frames(9, B, hw[0], hw[1]).astype(np.float64)
    loc = M.Localizer50(lp, crop, train=True, rng=np.random.RandomState(0))
    rois, points = loc.forward(frames)
    l_dir, g_dir = C.direction_loss(points, hw)
    l_oob, g_oob = C.out_of_image_loss(points)
    g_rois = np.random.RandomState(1).standard_normal(rois.shape)
    grads = {}
    loc.backward(g_rois, g_dir + g_oob, grads)

    tl = T.to_torch(lp, torch.float64)
    t_rois, t_points, t_theta = T.localizer50(tl, torch.tensor(frames), crop)
    d, o = T.regularisers(t_points, hw)
    loss = (t_rois * torch.tensor(g_rois)).sum() + d + o
    keys = [k for k, v in tl.items() if v.requires_grad]
    tg = dict(zip(keys, torch.autograd.grad(loss, [tl[k] for k in keys], allow_unused=True)))
    np.testing.assert_allclose(loc.theta, t_theta.detach().numpy(), rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(points, t_points.detach().numpy(), rtol=1e-8, atol=1e-10)
    checked = 0
    for k, g in grads.items():
        if tg.get(k) is None or k == 'feature_extractor/conv1/b':     # analytically zero (BN follows): rounding noise
            continue
        ref = tg[k].numpy()
        assert np.abs(g - ref).max() < 1e-6 * (np.abs(ref).max() + 1e-30) + 1e-12, k
        checked += 1
    assert checked >= 150
