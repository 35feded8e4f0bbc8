"""Golden vectors (tests/golden/step_b2_64.npz, made by tests/golden/make_golden.py from the fp64 oracle):
CPU: the oracle still reproduces them; GPU: the HIP path matches them without running the oracle."""
import os

import numpy as np
import pytest

from tests.golden import make_golden as G

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'step_b2_64.npz')


def test_oracle_reproduces_golden():
    gold = np.load(GOLD)
    out = G.run(np.float64)
    for k in gold.files:
        np.testing.assert_allclose(out[k], gold[k], rtol=1e-9, atol=1e-12, err_msg=k)
    out32 = G.run(np.float32)
    for k in ('theta', 'points', 'y_fake', 'y_real', 'bboxes_px'):
        scale = 64.0 if k == 'bboxes_px' else 1.0
        np.testing.assert_allclose(out32[k], gold[k], rtol=0, atol=1e-4 * scale, err_msg=k)


@pytest.mark.gpu
def test_hip_step_matches_golden():
    import torch
    import loans_amd
    from loans_amd.runtime import training
    gold = np.load(GOLD)
    lp, dp, frames, real, labels = G.setup(np.float32)
    np.random.seed(0)
    loc, dis = loans_amd.SheepLocalizer(G.CROP), loans_amd.ResnetAssessor()
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()     # noqa: E731
    with loans_amd.using_config('enable_backprop', False):
        dis(d(real))                                                      # materialise l4
    loc.load_state_dict_chainer(lp)
    dis.load_state_dict_chainer(dp)
    og = loans_amd.Adam(alpha=1e-3, amsgrad=True).setup(loc)
    od = loans_amd.Adam(alpha=1e-3, amsgrad=True).setup(dis)
    upd = loans_amd.SheepAssessor(
        models=[loc, dis], iterator={'main': training.DeviceBatchIterator([d(frames)]),
                                     'real': training.DeviceBatchIterator([(d(real), d(labels))])},
        optimizer={'opt_gen': og, 'opt_dis': od}, converter=training.identity_converter, device=0)
    x_fake, bboxes = loc(d(frames))                   # forward only, to compare outputs before the update
    y_fake = dis(x_fake)
    np.testing.assert_allclose(loc.last_transform_params.data.cpu().numpy(), gold['theta'], atol=1e-4, rtol=0)
    np.testing.assert_allclose(bboxes.data.cpu().numpy(), gold['points'], atol=1e-4, rtol=0)
    np.testing.assert_allclose(y_fake.data.cpu().numpy(), gold['y_fake'], atol=1e-4, rtol=0)
    px = loc.scale_bboxes(loc.extract_corners(bboxes), loans_amd.Size(64, 64)).cpu().numpy()
    np.testing.assert_allclose(px, gold['bboxes_px'], atol=1e-4 * 64, rtol=0)
    x_fake.unchain_backward(); bboxes.unchain_backward()
    # that forward moved the BN running statistics once; restore them, then take the real step
    loc.load_state_dict_chainer(lp)
    upd.update()
    obs = loans_amd.reporter.observation
    np.testing.assert_allclose(float(obs['loss_localizer']), gold['loss_localizer'], rtol=1e-4)
    np.testing.assert_allclose(float(obs['loss_dis']), gold['loss_dis'], rtol=1e-4)
    st = loc.state_dict_chainer()
    np.testing.assert_allclose(st['feature_extractor/bn1/avg_mean'], gold['new_bn1_avg_mean'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(st['param_predictor/b'], gold['new_param_predictor_b'], atol=2e-4, rtol=0)
