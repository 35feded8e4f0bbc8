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
    out.update(G.run_trajectory(np.float64))
    for k in out:
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


def test_oracle_reproduces_golden_224():
    """the B = 8, 3 x 224 x 224 forward vector in the oracle's fp32 arm: BASELINE's 1e-4 against the stored
    fp64 values"""
    gold = np.load(GOLD)
    out = G.run224(np.float32)
    np.testing.assert_allclose(out['theta224'], gold['theta224'], rtol=0, atol=1e-4)
    np.testing.assert_allclose(out['y_fake224'], gold['y_fake224'], rtol=0, atol=1e-4)
    np.testing.assert_allclose(out['corners224'], gold['corners224'], rtol=0, atol=1e-4 * 224)
    np.testing.assert_allclose(out['points224_corner'], gold['points224_corner'], rtol=0, atol=1e-4)


@pytest.mark.gpu
def test_hip_forward_matches_golden_224():
    """HIP forward at B = 8, 3 x 224 x 224, crop 75 x 75 against the stored vector -- no oracle at run time"""
    import torch
    import loans_amd
    gold = np.load(GOLD)
    lp, dp, frames = G.setup224(np.float32)
    np.random.seed(0)
    loc, dis = loans_amd.SheepLocalizer(G.CROP224), loans_amd.ResnetAssessor()
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()     # noqa: E731
    with loans_amd.using_config('enable_backprop', False):
        dis(torch.zeros(2, 3, 75, 75, device='cuda'))                     # materialise l4
    loc.load_state_dict_chainer(lp)
    dis.load_state_dict_chainer(dp)
    rois, points = loc(d(frames))
    y = dis(rois)
    np.testing.assert_allclose(loc.last_transform_params.data.cpu().numpy(), gold['theta224'], rtol=0, atol=1e-4)
    np.testing.assert_allclose(y.data.cpu().numpy(), gold['y_fake224'], rtol=0, atol=1e-4)
    px = loc.scale_bboxes(loc.extract_corners(points), loans_amd.Size(224, 224)).cpu().numpy()
    np.testing.assert_allclose(px, gold['corners224'], rtol=0, atol=1e-4 * 224)
    p = points.data.cpu().numpy()
    np.testing.assert_allclose(p[:, :, [0, 0, -1, -1], [0, -1, 0, -1]], gold['points224_corner'], rtol=0, atol=1e-4)
    np.testing.assert_allclose(rois.data.cpu().numpy().mean(axis=(1, 2, 3)), gold['rois224_mean'], rtol=0, atol=1e-4)


@pytest.mark.gpu
def test_hip_trajectory_matches_golden(deterministic_forward):
    """three consecutive joint steps of the B = 2 case: losses against the stored fp64 trajectory (Adam's sign-like first
    steps amplify rounding, so the bound is the one the fp32 oracle itself needs, see test_three_iteration_trajectory)"""
    import torch
    import loans_amd
    from loans_amd.runtime import training
    gold = np.load(GOLD)
    lp, dp, frames, real, labels = G.setup(np.float32)
    np.random.seed(0)
    loc, dis = loans_amd.SheepLocalizer(G.CROP), loans_amd.ResnetAssessor()
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()     # noqa: E731
    with loans_amd.using_config('enable_backprop', False):
        dis(d(real))
    loc.load_state_dict_chainer(lp)
    dis.load_state_dict_chainer(dp)
    upd = loans_amd.SheepAssessor(
        models=[loc, dis], iterator={'main': training.DeviceBatchIterator([d(frames)]),
                                     'real': training.DeviceBatchIterator([(d(real), d(labels))])},
        optimizer={'opt_gen': loans_amd.Adam(alpha=1e-3, amsgrad=True).setup(loc),
                   'opt_dis': loans_amd.Adam(alpha=1e-3, amsgrad=True).setup(dis)},
        converter=training.identity_converter, device=0)
    got = []
    for _ in range(3):
        upd.update()
        obs = loans_amd.reporter.observation
        got.append((float(obs['loss_localizer']), float(obs['loss_dis'])))
    got, ref = np.array(got), gold['traj_losses']
    np.testing.assert_allclose(got[0], ref[0], rtol=1e-4)
    # later steps: the bound the fp32 ORACLE itself needs against the fp64 one (stored beside the fp64 trajectory)
    full = np.load(FULL)
    np.testing.assert_allclose(full['full_losses'], ref, rtol=1e-12)
    tol = np.maximum(np.maximum(5 * np.abs(full['full_losses_f32'] - ref), 5e-4 * np.abs(ref)), 1e-5)
    assert (np.abs(got - ref) <= tol).all(), (got, ref, tol)
    np.testing.assert_allclose(float(dis.state_dict_chainer()['l4/W'].sum()), float(gold['traj_l4_W_sum']), rtol=5e-2, atol=5e-2)


# --------------------------------------------------------------------------------------------------------------------------
# SURVEY 8c in full: every parameter gradient and every post-update parameter of three consecutive steps (digests: L2 norm,
# sum, 48 seeded entries per tensor -- tests/golden/steps_full_b2_64.npz).  PARITY UNPINNED: made by this repo's fp64 oracle.
# --------------------------------------------------------------------------------------------------------------------------
FULL = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'steps_full_b2_64.npz')


def test_oracle_reproduces_full_steps():
    gold = np.load(FULL)
    out = G.run_full_steps(np.float64)
    assert list(out['full_loc_keys']) == list(gold['full_loc_keys']) and len(out['full_loc_keys']) == 66 + 30
    assert list(out['full_dis_keys']) == list(gold['full_dis_keys']) and len(out['full_dis_keys']) == 11
    for k in out:
        if k.endswith('_keys'):
            continue
        np.testing.assert_allclose(out[k], gold[k], rtol=1e-8, atol=1e-11, err_msg=k)


@pytest.mark.gpu
def test_hip_full_steps_match_golden(deterministic_forward):
    """ALL 66 + 11 parameter gradients and post-update parameters of the HIP path, three steps, against the stored digests.
    Gradients are read through Chainer-style optimiser hooks (called once per update, before the parameters move)."""
    import torch
    import loans_amd
    from loans_amd.runtime import training
    gold = np.load(FULL)
    lp, dp, frames, real, labels = G.setup(np.float32)
    np.random.seed(0)
    loc, dis = loans_amd.SheepLocalizer(G.CROP), loans_amd.ResnetAssessor()
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()     # noqa: E731
    with loans_amd.using_config('enable_backprop', False):
        dis(d(real))
    loc.load_state_dict_chainer(lp)
    dis.load_state_dict_chainer(dp)
    og = loans_amd.Adam(alpha=1e-3, amsgrad=True).setup(loc)
    od = loans_amd.Adam(alpha=1e-3, amsgrad=True).setup(dis)
    seen = {'loc': [], 'dis': []}
    og.add_hook(lambda opt: seen['loc'].append({k[1:]: p.grad_logical().copy() for k, p in opt.target.namedparams()}), 'capture')
    od.add_hook(lambda opt: seen['dis'].append({k[1:]: p.grad_logical().copy() for k, p in opt.target.namedparams()}), 'capture')
    upd = loans_amd.SheepAssessor(
        models=[loc, dis], iterator={'main': training.DeviceBatchIterator([d(frames)]),
                                     'real': training.DeviceBatchIterator([(d(real), d(labels))])},
        optimizer={'opt_gen': og, 'opt_dis': od}, converter=training.identity_converter, device=0)
    lkeys, dkeys = list(gold['full_loc_keys']), list(gold['full_dis_keys'])
    ref_l, f32_l = gold['full_losses'], gold['full_losses_f32']
    for it in range(3):
        upd.update()
        obs = loans_amd.reporter.observation
        got = np.array([float(obs['loss_localizer']), float(obs['loss_dis'])])
        tol = np.maximum(np.maximum(5 * np.abs(f32_l[it] - ref_l[it]), 5e-4 * np.abs(ref_l[it])), 1e-5)
        assert (np.abs(got - ref_l[it]) <= tol).all(), (it, got, ref_l[it], tol)
        np.testing.assert_allclose(loc.last_transform_params.data.cpu().numpy(), gold['full_theta_%d' % it], atol=2e-4 * (it + 1))
        states = {'loc': loc.state_dict_chainer(), 'dis': dis.state_dict_chainer()}
        for which, keys in (('loc', lkeys), ('dis', dkeys)):
            grads, ref_g, ref_p = seen[which][it], gold['full_%s_grad_%d' % (which, it)], gold['full_%s_param_%d' % (which, it)]
            n_off = n_tot = 0
            for row, k in enumerate(keys):
                if k.startswith(('res6', 'res7')):                         # 64 px frames: outside the active arena prefix
                    assert ref_g[row, 0] == 0 and not grads[k].any(), k
                    np.testing.assert_array_equal(G.digest(k, states[which][k])[2:], G.digest(k, (lp if which == 'loc' else dp)[k])[2:])
                    continue
                dg = G.digest(k, grads[k])
                if it == 0 and k != 'feature_extractor/conv1/b':           # conv1/b: analytically zero (a BN follows)
                    # step 1 starts from identical parameters: every gradient to 1e-3 of its tensor's scale
                    assert abs(dg[0] - ref_g[row, 0]) <= 1e-3 * ref_g[row, 0] + 1e-12, (k, dg[0], ref_g[row, 0])
                    scale = np.abs(ref_g[row, 2:]).max() + ref_g[row, 0] / np.sqrt(grads[k].size)
                    assert np.abs(dg[2:] - ref_g[row, 2:]).max() <= 2e-3 * scale, k
                dp_ = G.digest(k, states[which][k])
                off = np.abs(dp_[2:] - ref_p[row, 2:])
                # Adam's first steps are sign-like: an entry whose gradient is rounding noise may move the other way, never by
                # more than ~2 lr per step
                assert off.max() < 2.1e-3 * (it + 1), (it, k, off.max())
                n_off += int((off > 5e-5 * (it + 1)).sum())
                n_tot += off.size
            assert n_off <= max(2, 0.01 * n_tot * (it + 1)), (it, which, n_off, n_tot)
    assert len(seen['loc']) == 3 and len(seen['dis']) == 3


CONFIG1 = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'config1_b8_224.npz')


def test_oracle_reproduces_config1_first_iteration():
    """BASELINE configs[0] (batch 8, 3 x 224 x 224, crop 75 x 75): the stored 10-iteration trajectory starts where the oracle
    starts today (one fp64 iteration; the full ten take minutes)"""
    gold = np.load(CONFIG1)
    assert gold['losses'].shape == (10, 2) and gold['theta'].shape == (10, 8, 6)
    out = G.run_config1(np.float64, iterations=1)
    np.testing.assert_allclose(out['losses'][0], gold['losses'][0], rtol=1e-9)
    np.testing.assert_allclose(out['theta'][0], gold['theta'][0], rtol=0, atol=1e-12)
    # a fresh localizer: theta is the initial bias for every frame (SURVEY 8c KAT 1), and only it has moved after step 1
    np.testing.assert_allclose(gold['theta'][0], np.tile([[.8, 0, 0, 0, .8, 0]], (8, 1)), atol=1e-7)
