#!/usr/bin/env python
"""Generates tests/golden/configs1_b256_224.npz: the CPU oracle's joint step at BASELINE configs[1]'s OWN size
(256 x 3 x 224 x 224 frames, crop 75 x 75; tests/golden/fullsize_case.py) in float64, and in float32 for the drift bound.

Provenance as for make_golden.py: the reference holds no vectors and Chainer 4.1 is not installable, so these are outputs
of THIS repo's oracle (oracle/model.py) -- a regression pin and a run-time check for the GPU test that needs no oracle
run at that size (which takes minutes and tens of GB).

The oracle keeps every convolution's im2col matrix for its backward (as Chainer's CPU path does): ~50 GB in fp32 at this
size.  Here conv2d_fwd / conv2d_bwd are wrapped so that the forward walks the batch in chunks and hands the INPUT (a
reference, no copy) through the `col` slot, and the backward rebuilds each chunk's im2col from it: same arithmetic per
sample, weight gradients summed chunk by chunk.

    python tests/golden/make_fullsize_golden.py                 (~30 min on 8 cores, ~35 GB)
    python tests/golden/make_fullsize_golden.py --batch 128     tests/golden/configs3_b128_224.npz: one rank's shard of configs[3]
"""
import os
import resource
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import chainer_ops as C           # noqa: E402
from oracle import model as M                 # noqa: E402
from tests.golden import fullsize_case as K   # noqa: E402

CHUNK = 16
_fwd, _bwd = C.conv2d_fwd, C.conv2d_bwd


def lean_conv2d_fwd(x, W, b, stride, pad):
    ys = [_fwd(x[i:i + CHUNK], W, b, stride, pad)[0] for i in range(0, len(x), CHUNK)]
    return np.concatenate(ys), (x,)


def lean_conv2d_bwd(x_shape, col, W, gy, stride, pad, has_bias, need_gx=True):
    x = col[0]
    kh, kw = W.shape[2:]
    gW, gb, gxs = 0, (0 if has_bias else None), []
    for i in range(0, len(x), CHUNK):
        xc = x[i:i + CHUNK]
        cc = C.im2col(xc, kh, kw, stride, stride, pad, pad)
        gx, w_, b_ = _bwd(xc.shape, cc, W, gy[i:i + CHUNK], stride, pad, has_bias, need_gx)
        gW = gW + w_
        if has_bias:
            gb = gb + b_
        gxs.append(gx)
    return (np.concatenate(gxs) if need_gx else None), gW, gb


BATCH = K.B


def run(dtype, log):
    loc, dis = K.build_models()
    lp = M.cast_params(loc.state_dict_chainer(), dtype)
    dp = M.cast_params(dis.state_dict_chainer(), dtype)
    frames, real, labels = K.build_inputs(BATCH)
    t0 = time.time()
    res = M.update_core(lp, dp, M.AdamAMSGrad(lp), M.AdamAMSGrad(dp), frames.astype(dtype), real.astype(dtype),
                        labels.astype(dtype), K.CROP, rng=np.random.RandomState(0), return_grads=True)
    log('%s step: %.0f s, peak RSS %.1f GB, losses %.6f %.6f' % (
        np.dtype(dtype).name, time.time() - t0, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6,
        res['loss_localizer'], res['loss_dis']))
    lkeys = sorted(k for k in lp if M.is_trainable(k) and not k.startswith(('res6', 'res7')))
    dkeys = sorted(k for k in dp if M.is_trainable(k))
    out = dict(theta=res['theta'], corners=M.Localizer(lp, K.CROP).corners_px(res['points'], (K.HW, K.HW)),
               y_fake=res['y_fake'], y_real=res['y_real'],
               loss_localizer=np.float64(res['loss_localizer']), loss_dis=np.float64(res['loss_dis']),
               loc_keys=np.array(lkeys), dis_keys=np.array(dkeys),
               loc_grad_norm=np.array([np.linalg.norm(np.asarray(res['loc_grads'][k], np.float64)) for k in lkeys]),
               dis_grad_norm=np.array([np.linalg.norm(np.asarray(res['dis_grads'][k], np.float64)) for k in dkeys]),
               loc_grad_sum=np.array([np.asarray(res['loc_grads'][k], np.float64).sum() for k in lkeys]))
    for bn in K.BN_KEYS:             # batch statistics, read back through one step's running averages (decay 0.9 from 0 / 1)
        out['avg_mean:' + bn] = lp[bn + '/avg_mean']
        out['avg_var:' + bn] = lp[bn + '/avg_var']
    return out


if __name__ == '__main__':
    resource.setrlimit(resource.RLIMIT_AS, (56 << 30, 56 << 30))      # a MemoryError, not the kernel's OOM killer
    C.conv2d_fwd, C.conv2d_bwd = lean_conv2d_fwd, lean_conv2d_bwd
    here = os.path.dirname(os.path.abspath(__file__))
    log = lambda s: print(s, flush=True)      # noqa: E731
    fixture = K.FIXTURE
    if sys.argv[1:3] == ['--batch', str(K.SHARD_B)]:           # one rank's shard of configs[3]
        BATCH, fixture = K.SHARD_B, K.SHARD_FIXTURE
    elif sys.argv[1:]:
        raise SystemExit('usage: make_fullsize_golden.py [--batch %d]' % K.SHARD_B)
    r32 = run(np.float32, log)
    out = {k + '_f32': v for k, v in r32.items() if not k.endswith('_keys')}
    try:
        r64 = run(np.float64, log)
    except MemoryError:
        log('float64 step does not fit: the fixture carries the float32 oracle only')
        r64 = {k: v for k, v in r32.items()}
        out['f64_missing'] = np.array(1)
    out.update(r64)
    path = os.path.join(here, fixture)
    np.savez_compressed(path, **out)
    log('wrote %s (%d bytes)' % (path, os.path.getsize(path)))
