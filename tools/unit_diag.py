#!/usr/bin/env python
"""Every residual unit of a localizer IN SITU against the fp64 oracle, teacher-forced (the unit's oracle twin gets the tensors
the HIP unit received): where along the chain a gradient discrepancy enters.  fp32 arm.  usage: unit_diag.py [B H W seed]"""
import contextlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import loans_amd                                   # noqa: E402
from loans_amd import ops                          # noqa: E402
from oracle import chainer_ops as C                # noqa: E402
from oracle import model as M                      # noqa: E402
from tests.gpu_util import dev, inputs, oracle_params, randomize_bn_and_predictor      # noqa: E402


def _l2(a, ref):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return float(np.linalg.norm(a - ref) / (np.linalg.norm(ref) + 1e-300))


def _ulp_profile(a, o):
    a, o = np.asarray(a, np.float64), np.asarray(o, np.float64)
    return (float(np.abs(a - o).max() / (np.abs(o).max() + 1e-300)),)


def _hip_units(loc):
    """the residual units of a localizer in execution order (sheep/resnet.py BasicA / BasicB, Chainer / chainercv bottlenecks)"""
    fe, out = loc.feature_extractor, []
    for st in (fe.res2, fe.res3, fe.res4, fe.res5, loc.res6, loc.res7):
        names = getattr(st, '_forward', None)
        out += [getattr(st, n) for n in names] if names else list(st.children())
    return out


def _nchw(t):
    return t.float().cpu().numpy().transpose(0, 3, 1, 2).astype(np.float64)


def _teacher_forced_units(loc_cls, oracle_cls, B, H, W, crop, seed):
    """Every conv / BN layer of the localizer IN SITU -- real weights, real activations and real gradients of a B x 3 x H x W
    step of the bf16 arm -- against the bf16-rounding oracle, unit by unit: each unit's oracle twin gets the tensors the HIP
    unit actually received (its bf16 input on the way up, its bf16 output gradient on the way down) and must reproduce what the
    HIP unit produced.  Free-running the two networks side by side instead compares little: a bf16 network amplifies a one-ulp
    difference by 2 - 3 x per residual unit (measured: 2e-5 after the stem, 4.5e-2 after 13 units), so after the first
    rounding that falls the other way the two runs decorrelate -- which says nothing about either."""
    from loans_amd.functions import blocks, global_average_pooling_2d, linear, reshape, rotation_dropout, spatial_transformer_grid
    from loans_amd.runtime.core import Variable
    np.random.seed(seed)
    loc = loc_cls(crop)
    randomize_bn_and_predictor(loc, np.random.RandomState(seed + 100))
    frames = inputs(seed + 2, B, H, W, crop)[0]
    loc.finalize(torch.device('cuda', 0))
    loc.arena.set_active(None if loc_cls is loans_amd.SheepLocalizer else 'feature_extractor/fc6')
    lp = oracle_params(loc, np.float64)
    key_of = {id(p): k[1:] for k, p in loc.namedparams()}
    fe = loc.feature_extractor
    report = []

    def param_errs(link_params, grads):
        errs = {}
        for p in link_params:
            k = key_of[id(p)]
            if k in grads and not k.endswith('conv1/b') and np.linalg.norm(grads[k]) > 0:
                errs[k] = _l2(p.grad_logical(), grads[k])
        return errs

    # ---- forward, HIP: the real chain; every unit's input is kept ----
    x = loc.prepare_images(dev(frames))
    stem_fn = lambda v: blocks.StemFunction(fe.conv1, fe.bn1)(v, fe.conv1.W, fe.conv1.b, fe.bn1.gamma, fe.bn1.beta)     # noqa: E731
    units = _hip_units(loc)
    h = stem_fn(x)
    ins = [h]
    for u in units:
        h = u(h)
        ins.append(h)
    feat = Variable(ins[-1].data, requires_grad=True)

    # ---- head (fp32 on both sides): GAP -> Linear -> rotation dropout -> grid -> the two regularisers ----
    pooled = global_average_pooling_2d(feat)
    theta = rotation_dropout(reshape(linear(pooled, loc.param_predictor.W, loc.param_predictor.b), (-1, 2, 3)), ratio=0.0)
    points = spatial_transformer_grid(theta, crop)
    size = loans_amd.Size(H, W)
    loss = loans_amd.DirectionLossCalculator(torch).calc_loss(points, size)
    loss = loss + loans_amd.OutOfImageLossCalculator(torch).calc_loss(points, size)
    loc.cleargrads()
    loss.backward()
    f64 = _nchw(feat.data)
    o_pooled = C.gap_fwd(f64)
    o_theta = C.linear_fwd(o_pooled, lp['param_predictor/W'], lp['param_predictor/b']).reshape(-1, 2, 3)
    mask = C.rotation_dropout_mask(o_theta, 0.0, True, np.random.RandomState(0))
    o_theta = o_theta * mask
    o_points, coords = C.st_grid_fwd(o_theta, crop)
    np.testing.assert_allclose(theta.data.cpu().numpy(), o_theta, atol=2e-5)
    g_pts = C.direction_loss(o_points, (H, W))[1] + C.out_of_image_loss(o_points)[1]
    assert np.abs(g_pts).max() > 0                                # the regularisers are active for this seed
    g_theta = C.st_grid_bwd(coords, g_pts) * mask
    g_pooled, gW, gb = C.linear_bwd(o_pooled, lp['param_predictor/W'], g_theta.reshape(-1, 6), True)
    g_feat = C.gap_bwd(f64.shape, g_pooled)
    head = {'param_predictor/W': _l2(loc.param_predictor.W.grad_logical(), gW),
            'param_predictor/b': _l2(loc.param_predictor.b.grad_logical(), gb),
            'd loss / d features': _l2(_nchw(feat.grad), g_feat)}
    report.append(('head', 0.0, (0.0, 0.0, 0.0, 0.0), head['d loss / d features'], head, 1 << 30))
    g = feat.grad

    # ---- the residual units, last to first: HIP unit on its real input with its real output gradient ----
    o_units = oracle_cls(lp, crop, train=True, rng=np.random.RandomState(0))._make_blocks(H)
    assert len(o_units) == len(units)
    for i in range(len(units) - 1, -1, -1):
        leaf = Variable(ins[i].data, requires_grad=True)
        out = units[i](leaf)
        assert torch.equal(out.data, ins[i + 1].data)
        out.grad = g
        loc.cleargrads()
        out.backward()
        ops.join_side_stream()
        with contextlib.nullcontext():
            o_out = o_units[i].fwd(_nchw(ins[i].data))
            grads = {}
            o_gx = o_units[i].bwd(_nchw(g), grads)
        a = _nchw(out.data)
        errs = param_errs(list(units[i].params()), grads)
        assert len(errs) >= 6
        report.append((key_of[id(next(iter(units[i].params())))].rsplit('/', 2)[0], _l2(a, o_out), _ulp_profile(a, o_out),
                       _l2(_nchw(leaf.grad), o_gx), errs, a.shape[0] * a.shape[2] * a.shape[3]))
        g = leaf.grad

    # ---- the stem: conv1 7x7/2 + bias -> bn1 -> relu -> max-pool; no input gradient ----
    pooled_hip = stem_fn(x)
    assert torch.equal(pooled_hip.data, ins[0].data)
    pooled_hip.grad = g
    loc.cleargrads()
    pooled_hip.backward()
    ops.join_side_stream()
    with contextlib.nullcontext():
        stem = M._ConvBN(lp, 'feature_extractor/conv1', 'feature_extractor/bn1', 2, 3, True)
        sr = M._q(C.relu(stem.fwd(M._q(C.prepare_images(frames.astype(np.float64))))))
        o_pool, idx = C.max_pool_fwd(sr, 3, 2, 0)
        grads = {}
        stem.bwd(C.max_pool_bwd(sr.shape, idx, _nchw(g), 3, 2, 0) * (sr > 0), grads, need_gx=False)
    a = _nchw(ins[0].data)
    errs = param_errs([fe.conv1.W, fe.bn1.gamma, fe.bn1.beta], grads)
    report.append(('stem', _l2(a, o_pool), _ulp_profile(a, o_pool), 0.0, errs, a.shape[0] * a.shape[2] * a.shape[3]))
    return report




if __name__ == '__main__':
    a = [int(v) for v in sys.argv[1:]]
    B, H, W = a[:3] if len(a) >= 3 else (3, 320, 304)
    seed = a[3] if len(a) > 3 else 41
    ops.SPLITK = False
    rep = _teacher_forced_units(loans_amd.SheepLocalizer, M.Localizer, B, H, W, (20, 28), seed)
    for name, e_out, prof, e_gx, errs, n in reversed(rep):
        worst = max(errs, key=errs.get)
        print('%-28s out L2 %.2e (max rel %.2e)  gx L2 %.2e  parameter gradients L2 max %.2e (%s)  samples/channel %d'
              % (name, e_out, prof[0], e_gx, errs[worst], worst.split('/', 1)[-1] if '/' in worst else worst, n))
