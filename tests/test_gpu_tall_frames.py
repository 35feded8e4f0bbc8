"""-m gpu: the configuration that TRAINS res6 / res7 (frames taller than 300 px, sheep/sheep_localizer.py:51-55) against the
fp64 oracle: every localizer gradient (res6 / res7 included) before Adam, then one whole `update_core` -- res6 / res7 move,
the arena's active prefix is the whole arena, Adam-AMSGrad lands where the oracle's does.

PARITY UNPINNED: the oracle is this repo's restatement of Chainer's arithmetic, the reference holds no vectors.
"""
import numpy as np
import pytest
import torch

import loans_amd
from oracle import model as M
from tests.gpu_util import build_pair, dev, inputs, oracle_params, rel_err
from tests.test_gpu_model import _updater

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures('kernel_selection')]


def test_res6_res7_gradients_and_update_parity(deterministic_forward):
    B, H, W, crop = 3, 320, 304, (20, 28)
    loc, dis = build_pair(41, crop)
    frames, real, labels = inputs(42, B, H, W, crop)
    with loans_amd.using_config('enable_backprop', False):
        dis(dev(real))                                        # materialises the lazy l4
    loc.finalize(torch.device('cuda', 0))
    lp, dp = oracle_params(loc, np.float64), oracle_params(dis, np.float64)
    lp32, dp32 = oracle_params(loc, np.float32), oracle_params(dis, np.float32)
    lp0 = {k: v.copy() for k, v in lp.items()}

    f64 = [a.astype(np.float64) for a in (frames, real, labels)]
    res = M.update_core(lp, dp, M.AdamAMSGrad(lp), M.AdamAMSGrad(dp), f64[0], f64[1], f64[2], crop,
                        rng=np.random.RandomState(0), return_grads=True)
    r32 = M.update_core(lp32, dp32, M.AdamAMSGrad(lp32), M.AdamAMSGrad(dp32), frames, real, labels, crop,
                        rng=np.random.RandomState(0), return_grads=True)
    assert any(k.startswith('res7/') for k in res['loc_grads']) and any(k.startswith('res6/') for k in res['loc_grads'])

    # ---- 1. the localizer chain's gradients, before Adam (the first half of update_core by hand) ----
    x_fake, bboxes = loc(dev(frames))
    assert loc.arena.active_numel == loc.arena.numel            # res6 and res7 are inside the active prefix
    y_fake = dis(x_fake)
    loss = loans_amd.functions.mean_squared_error(y_fake, torch.full((B, 1), 1.0, device='cuda'))
    size = loans_amd.Size(H, W)
    loss = loss + loans_amd.DirectionLossCalculator(torch).calc_loss(bboxes, size)
    loss = loss + loans_amd.OutOfImageLossCalculator(torch).calc_loss(bboxes, size)
    dis.disable_update()
    loc.cleargrads()
    loss.backward()
    dis.enable_update()
    np.testing.assert_allclose(float(loss.data), res['loss_localizer'], rtol=1e-4)
    # res7's BNs normalise over B x 3 x 3 = 27 samples per channel: ill-conditioned in fp32 for ANY implementation -- a last-bit
    # difference in the forward comes back 1e4 times larger in the gradients of every stage below.  How ill-conditioned, the
    # fp32 ORACLE's own distance from the fp64 one says (e32); the bound follows it where it exceeds the 1e-3 of the 64 x 64
    # test.  Per STAGE, not per tensor: e and e32 are two draws of the same rounding noise, and one tensor's e32 happening to
    # come out small says nothing about the conditioning of the 15-odd tensors of its stage (round 4: with another, equally
    # valid, tile for conv1 res4/1/bn2/beta read e = 1.4e-2 against its own e32 = 1.4e-3 while res4's largest e32 was higher).
    stage_of = lambda key: (key.split('/')[1] if key.startswith('/res') else key.split('/')[2] if 'res' in key else 'head')   # noqa: E731
    worst, worst32, errs = {}, {}, {}
    for key, p in loc.namedparams():
        ref = res['loc_grads'].get(key[1:])
        assert ref is not None, key                              # at this height every parameter has a gradient
        if key == '/feature_extractor/conv1/b':
            continue                                             # analytically zero (BN follows): rounding noise
        errs[key] = rel_err(p.grad_logical(), ref)
        e32 = rel_err(r32['loc_grads'][key[1:]], ref)            # the fp32 ORACLE's own distance from fp64 on this tensor
        worst[stage_of(key)] = max(worst.get(stage_of(key), 0.0), errs[key])
        worst32[stage_of(key)] = max(worst32.get(stage_of(key), 0.0), e32)
    print('fp32 oracle vs fp64 oracle, worst per stage:    ', {k: '%.2e' % v for k, v in sorted(worst32.items())})
    for key, e in errs.items():
        assert e < max(1e-3, 10 * worst32[stage_of(key)]), (key, e, worst32[stage_of(key)])
    print('worst relative gradient error per stage:', {k: '%.2e' % v for k, v in sorted(worst.items())})
    assert 'res6' in worst and 'res7' in worst

    # ---- 2. one whole update_core from the same initial state ----
    for _, link, n in loc.namedpersistents():                    # undo the running-statistics update of the pass above
        v = getattr(link, n)
        if torch.is_tensor(v):
            v.fill_(1.0 if n == 'avg_var' else 0.0)
    upd = _updater(loc, dis, frames, real, labels)
    upd.update()
    obs = loans_amd.reporter.observation
    np.testing.assert_allclose(float(obs['loss_localizer']), res['loss_localizer'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(float(obs['loss_dis']), res['loss_dis'], rtol=1e-4, atol=1e-6)
    new_loc = loc.state_dict_chainer()
    moved = {'res6': 0.0, 'res7': 0.0}
    for key in lp:
        if not M.is_trainable(key) or key == 'feature_extractor/conv1/b':
            continue
        d = np.abs(new_loc[key] - lp[key])
        # Adam is sign-like on step 1 (|update| ~ lr whatever the gradient's size): entries whose gradient is rounding noise may
        # flip sign -- never more than ~2 lr apart, off by more than 5 % of lr on at most 0.2 % of the entries (the criterion of
        # test_update_core_gradients_and_parameters_parity)
        assert d.max() < 2.1e-3, key
        # ... or as many as the fp32 ORACLE itself flips against the fp64 one on this tensor
        # (lp32 was stepped by r32 above); a 64-entry BN vector may hold one such entry
        n32 = int(np.sum(np.abs(lp32[key].astype(np.float64) - lp[key]) > 5e-5))
        assert np.sum(d > 5e-5) <= max(2e-3 * d.size, 3 * n32, 2), (key, int(np.sum(d > 5e-5)), n32, d.size)
        for st in moved:
            if key.startswith(st + '/'):
                moved[st] = max(moved[st], float(np.abs(new_loc[key] - lp0[key]).max()))
    assert moved['res6'] > 5e-4 and moved['res7'] > 5e-4, moved   # both cold stages were trained (~lr per entry)
    new_dis = dis.state_dict_chainer()
    for key in dp:
        d = np.abs(new_dis[key] - dp[key])
        assert np.mean(d > 5e-5) < 2e-3, (key, np.mean(d > 5e-5))
    # BN running statistics of the cold stages moved as the oracle's did
    for k in ('res6/0/bn1/avg_mean', 'res7/1/bn2/avg_var'):
        np.testing.assert_allclose(new_loc[k], lp[k], rtol=1e-3, atol=1e-5)
