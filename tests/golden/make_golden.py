#!/usr/bin/env python
"""Generates tests/golden/step_b2_64.npz: expected outputs of ONE joint LoANs step on seeded inputs, the losses of the
first three steps, and one B = 8, 224 x 224 forward vector.

Provenance: the reference (Bartzi/loans) has no tests / golden vectors and its arithmetic (Chainer
4.1.0 / CuPy) is not installable here, so these vectors come from THIS repo's CPU oracle
(oracle/model.py, float64 arm), which tests/test_oracle_kat.py pins to source-derived known answers
and tests/test_oracle_vs_torch.py cross-checks against an independent torch-CPU autograd
composition.  They guard the oracle against regressions (CPU test) and give the HIP path a check that
does not need the oracle at run time (GPU test).  Inputs and weights are regenerated from seeds, only
the expected outputs are stored.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from loans_amd.datasets import synthetic   # noqa: E402
from oracle import model as M              # noqa: E402

CROP = (16, 16)


def setup(dtype=np.float64):
    rng = np.random.RandomState(2024)
    lp = M.cast_params(M.init_localizer_params(rng, predictor_w_std=5e-3), dtype)
    dp = M.cast_params(M.init_assessor_params(rng, CROP), dtype)
    frames = synthetic.make_frames(77, 2, 64, 64).astype(dtype)
    real, labels = synthetic.make_assessor_batch(78, 2, CROP[0], CROP[1], src=64)
    return lp, dp, frames, real.astype(dtype), labels.astype(dtype)


def run(dtype=np.float64):
    lp, dp, frames, real, labels = setup(dtype)
    res = M.update_core(lp, dp, M.AdamAMSGrad(lp), M.AdamAMSGrad(dp), frames, real, labels, CROP,
                        rng=np.random.RandomState(0), return_grads=True)
    loc = M.Localizer(lp, CROP)
    out = dict(theta=res['theta'], points=res['points'], y_fake=res['y_fake'], y_real=res['y_real'],
               loss_localizer=np.float64(res['loss_localizer']), loss_dis=np.float64(res['loss_dis']),
               bboxes_px=loc.corners_px(res['points'], (64, 64)),
               g_param_predictor_b=res['loc_grads']['param_predictor/b'],
               g_res3_conv3_W_sum=np.float64(res['loc_grads']['feature_extractor/res3/0/conv3/W'].sum()),
               g_r1_c1_W_absmax=np.float64(np.abs(res['dis_grads']['r1/c1/W']).max()),
               new_param_predictor_b=lp['param_predictor/b'], new_bn1_avg_mean=lp['feature_extractor/bn1/avg_mean'])
    return out


CROP224 = (75, 75)


def setup224(dtype=np.float64):
    """one B = 8, 3 x 224 x 224 forward vector with a non-zero seeded param_predictor.W."""
    rng = np.random.RandomState(224)
    lp = M.cast_params(M.init_localizer_params(rng, predictor_w_std=2e-2), dtype)
    dp = M.cast_params(M.init_assessor_params(rng, CROP224), dtype)
    frames = synthetic.make_frames(79, 8, 224, 224).astype(dtype)
    return lp, dp, frames


def run224(dtype=np.float64):
    lp, dp, frames = setup224(dtype)
    loc = M.Localizer(lp, CROP224, train=True, rng=np.random.RandomState(0))
    rois, points = loc.forward(frames)
    y = M.Assessor(dp).forward(rois)
    return dict(theta224=loc.theta, corners224=loc.corners_px(points, (224, 224)), y_fake224=y,
                rois224_mean=rois.mean(axis=(1, 2, 3)), points224_corner=points[:, :, [0, 0, -1, -1], [0, -1, 0, -1]])


def run_trajectory(dtype=np.float64, iterations=3):
    """losses and parameter check-sums of 3 consecutive joint steps of the B = 2 case."""
    lp, dp, frames, real, labels = setup(dtype)
    og, od = M.AdamAMSGrad(lp), M.AdamAMSGrad(dp)
    losses = []
    for _ in range(iterations):
        r = M.update_core(lp, dp, og, od, frames, real, labels, CROP, rng=np.random.RandomState(0))
        losses.append((r['loss_localizer'], r['loss_dis']))
    return dict(traj_losses=np.array(losses), traj_param_predictor_b=lp['param_predictor/b'],
                traj_l4_W_sum=np.float64(dp['l4/W'].sum()), traj_bn1_avg_var=lp['feature_extractor/bn1/avg_var'])


if __name__ == '__main__':
    out = run()
    out.update(run224())
    out.update(run_trajectory())
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'step_b2_64.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, {k: np.asarray(v).shape for k, v in out.items()})
