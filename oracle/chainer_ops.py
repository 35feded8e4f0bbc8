"""CPU oracle: NumPy restatement of the Chainer 4.1.0 operators on the LoANs hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``loans_amd/`` may import this package;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and there only as the checker / the timed CPU baseline.

PARITY UNPINNED: the reference (/root/reference, Bartzi/loans) ships no tests,
golden vectors or fixtures, and its arithmetic lives in third-party packages
that are absent from the tree and not installable here:
``chainer==4.1.0``, ``cupy==4.1.0``, ``chainercv==0.9.0``
(reference ``requirements.txt:1-3``).  Every function below restates the
*published* algorithm of that Chainer release for the NumPy (CPU) code path and
cites the reference call site that uses it.  The restatement is pinned only by
(a) analytic known-answer tests derived from reference source
(tests/test_oracle_kat.py) and (b) an independent torch-CPU composition
(tests/test_oracle_vs_torch.py).

All functions take/return NCHW float arrays like Chainer does.  ``dtype`` is
whatever the inputs carry (float32 for the parity oracle, float64 for the
high-precision arm).
"""
import math

import numpy as np

# Chainer 4.1.0's CPU batch-normalisation adds eps to the batch variance *in
# place* before it updates the running variance, so the running variance
# carries ``adjust * eps`` (chainer/functions/normalization/batch_normalization.py,
# "var += self.eps" ahead of "self.running_var += (1 - decay) * adjust * var").
RUNNING_VAR_INCLUDES_EPS = True

BN_EPS = 2e-5      # chainer.links.BatchNormalization default (sheep/resnet.py:44)
BN_DECAY = 0.9     # chainer.links.BatchNormalization default


# --------------------------------------------------------------------------- #
# preprocessing: sheep/sheep_localizer.py:45,72-82 -> chainer resnet.prepare
# --------------------------------------------------------------------------- #
RESNET_MEAN_BGR = np.array([103.063, 115.903, 123.152], dtype=np.float32)


def prepare_images(images):
    """``prepare_images(images.copy() * 255)`` (sheep_localizer.py:45,72-82).

    chainer.links.model.vision.resnet.prepare(image, size=None):
    ``Image.fromarray(image.astype(uint8))`` (truncation toward zero), RGB->BGR,
    minus the BGR mean, CHW.  Output is always float32 in Chainer; we keep the
    caller's dtype for the fp64 arm but quantise in float32 exactly as the
    reference does (the f32 multiply happens *before* the uint8 cast).
    """
    x = (images.astype(np.float32) * np.float32(255)).astype(np.uint8)
    x = x.astype(np.float32)[:, ::-1, :, :]                       # RGB -> BGR
    x = x - RESNET_MEAN_BGR.reshape(1, 3, 1, 1)
    return x.astype(images.dtype, copy=False)


# --------------------------------------------------------------------------- #
# convolution: L.Convolution2D -> F.convolution_2d CPU path (im2col + tensordot)
#   call sites: sheep/resnet.py:43,128-133,151-153 ; common/net.py:15-17,37-39,59-60
# --------------------------------------------------------------------------- #
def conv_outsize(size, k, s, p, cover_all=False):
    if cover_all:
        return (size + p * 2 - k + s - 1) // s + 1
    return (size + p * 2 - k) // s + 1


def im2col(x, kh, kw, sy, sx, ph, pw, pval=0.0, cover_all=False):
    n, c, h, w = x.shape
    oh = conv_outsize(h, kh, sy, ph, cover_all)
    ow = conv_outsize(w, kw, sx, pw, cover_all)
    img = np.pad(x, ((0, 0), (0, 0), (ph, ph + sy - 1), (pw, pw + sx - 1)),
                 mode='constant', constant_values=(pval,))
    col = np.ndarray((n, c, kh, kw, oh, ow), dtype=x.dtype)
    for j in range(kh):
        jlim = j + sy * oh
        for i in range(kw):
            ilim = i + sx * ow
            col[:, :, j, i, :, :] = img[:, :, j:jlim:sy, i:ilim:sx]
    return col


def col2im(col, sy, sx, ph, pw, h, w):
    n, c, kh, kw, oh, ow = col.shape
    img = np.zeros((n, c, h + 2 * ph + sy - 1, w + 2 * pw + sx - 1), dtype=col.dtype)
    for j in range(kh):
        jlim = j + sy * oh
        for i in range(kw):
            ilim = i + sx * ow
            img[:, :, j:jlim:sy, i:ilim:sx] += col[:, :, j, i]
    return img[:, :, ph:h + ph, pw:w + pw]


def conv2d_fwd(x, W, b, stride, pad):
    kh, kw = W.shape[2:]
    col = im2col(x, kh, kw, stride, stride, pad, pad)
    y = np.tensordot(col, W, ((1, 2, 3), (1, 2, 3))).astype(x.dtype, copy=False)
    if b is not None:
        y += b
    return np.ascontiguousarray(np.rollaxis(y, 3, 1)), col


def conv2d_bwd(x_shape, col, W, gy, stride, pad, has_bias, need_gx=True):
    """Returns (gx, gW, gb).  ``col`` is the im2col of the forward input."""
    h, w = x_shape[2:]
    gW = np.tensordot(gy, col, ((0, 2, 3), (0, 4, 5))).astype(W.dtype, copy=False)
    gb = gy.sum(axis=(0, 2, 3)) if has_bias else None
    gx = None
    if need_gx:
        gcol = np.tensordot(W, gy, (0, 1)).astype(gy.dtype, copy=False)
        gcol = np.rollaxis(gcol, 3)
        gx = col2im(gcol, stride, stride, pad, pad, h, w)
    return gx, gW, gb


# --------------------------------------------------------------------------- #
# batch normalisation: L.BatchNormalization (sheep/resnet.py:44,129-134,152-154)
# --------------------------------------------------------------------------- #
def round_bf16(a):
    """Round to the nearest bfloat16 (ties to even), returned in a's own dtype: what one store of the HIP bf16-storage arm
    does to a tensor.  Only used by oracle.model.emulate_bf16_storage (the oracle itself is fp32 / fp64)."""
    a = np.asarray(a)
    f = np.ascontiguousarray(a, dtype=np.float32)
    u = f.view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)).view(np.float32)
    r = np.where(np.isfinite(f), r, f)
    return r.astype(a.dtype, copy=False)


def bn_fwd_train(x, gamma, beta, running_mean, running_var, eps=BN_EPS, decay=BN_DECAY, x_apply=None):
    """Training-mode forward; updates running stats in place.  Returns (y, ctx).
    ``x_apply`` (bf16-storage emulation only): the statistics come from ``x`` (the conv's fp32 accumulators), the
    normalisation is applied to ``x_apply`` (the tensor as it was stored)."""
    axis = (0, 2, 3)
    ex = (None, slice(None), None, None)
    mean = x.mean(axis=axis)
    var = x.var(axis=axis)                       # biased
    var_eps = var + eps
    inv_std = var_eps ** (-0.5)
    x_hat = ((x if x_apply is None else x_apply) - mean[ex]) * inv_std[ex]
    y = gamma[ex] * x_hat
    y += beta[ex]
    m = x.size // gamma.size
    adjust = m / max(m - 1.0, 1.0)
    running_mean *= decay
    running_mean += (1 - decay) * mean
    running_var *= decay
    running_var += (1 - decay) * adjust * (var_eps if RUNNING_VAR_INCLUDES_EPS else var)
    return y, (x_hat, inv_std)


def bn_fwd_test(x, gamma, beta, running_mean, running_var, eps=BN_EPS):
    ex = (None, slice(None), None, None)
    inv_std = (running_var + eps) ** (-0.5)
    return gamma[ex] * ((x - running_mean[ex]) * inv_std[ex]) + beta[ex]


def bn_bwd(ctx, gamma, gy):
    """Standard two-reduction backward (chainer BatchNormalizationGrad)."""
    x_hat, inv_std = ctx
    axis = (0, 2, 3)
    ex = (None, slice(None), None, None)
    m = gy.size // gamma.size
    gbeta = gy.sum(axis=axis)
    ggamma = (gy * x_hat).sum(axis=axis)
    gx = (gamma * inv_std)[ex] * (gy - (x_hat * ggamma[ex] + gbeta[ex]) / m)
    return gx, ggamma, gbeta


def bn_bwd_test(gamma, running_var, gy, eps=BN_EPS):
    """Fixed-statistics backward (only the data gradient matters)."""
    ex = (None, slice(None), None, None)
    inv_std = (running_var + eps) ** (-0.5)
    return (gamma * inv_std)[ex] * gy


# --------------------------------------------------------------------------- #
# relu / max pooling (sheep/resnet.py:73 : max_pooling_2d(relu(h), 3, stride=2))
# --------------------------------------------------------------------------- #
def relu(x):
    return np.maximum(x, 0)


def max_pool_fwd(x, k=3, stride=2, pad=0):
    """cover_all=True (Chainer default); padding/overhang filled with -inf."""
    col = im2col(x, k, k, stride, stride, pad, pad, pval=-float('inf'), cover_all=True)
    n, c, kh, kw, oh, ow = col.shape
    col = col.reshape(n, c, kh * kw, oh, ow)
    indexes = col.argmax(axis=2)                 # first maximum in (kh,kw) order
    y = col.max(axis=2)
    return y, indexes


def max_pool_bwd(x_shape, indexes, gy, k=3, stride=2, pad=0):
    n, c, oh, ow = gy.shape
    h, w = x_shape[2:]
    gcol = np.zeros((n * c * oh * ow * k * k), dtype=gy.dtype)
    idx = indexes.ravel() + np.arange(0, indexes.size * k * k, k * k)
    gcol[idx] = gy.ravel()
    gcol = gcol.reshape(n, c, oh, ow, k, k)
    gcol = np.swapaxes(gcol, 2, 4)
    gcol = np.swapaxes(gcol, 3, 5)
    return col2im(gcol, stride, stride, pad, pad, h, w)


# --------------------------------------------------------------------------- #
# global average pooling + Linear (sheep_localizer.py:58,60 ; common/net.py:81,90)
# --------------------------------------------------------------------------- #
def gap_fwd(x):
    return x.mean(axis=(2, 3))


def gap_bwd(x_shape, gy):
    n, c, h, w = x_shape
    return np.broadcast_to((gy / (h * w))[:, :, None, None], x_shape).astype(gy.dtype)


def linear_fwd(x, W, b):
    x2 = x.reshape(len(x), -1)
    y = x2.dot(W.T).astype(x.dtype, copy=False)
    if b is not None:
        y += b
    return y


def linear_bwd(x, W, gy, has_bias):
    x2 = x.reshape(len(x), -1)
    gx = gy.dot(W).astype(x.dtype, copy=False).reshape(x.shape)
    gW = gy.T.dot(x2).astype(W.dtype, copy=False)
    gb = gy.sum(axis=0) if has_bias else None
    return gx, gW, gb


# --------------------------------------------------------------------------- #
# rotation dropout: functions/rotation_droput.py:26-48, called with ratio=0.0
# --------------------------------------------------------------------------- #
def rotation_dropout_mask(theta, ratio, train, rng=None):
    mask = np.ones_like(theta)
    if not train:
        flag = ratio                              # :33-35
    else:
        draw = (rng.random_sample(1) if rng is not None else np.random.rand(1))
        flag = float(draw[0] < ratio)             # :41-43
    mask[:, 0, 1] = flag
    mask[:, 1, 0] = flag
    return mask


# --------------------------------------------------------------------------- #
# spatial transformer (sheep_localizer.py:62-63)
# --------------------------------------------------------------------------- #
def st_grid_fwd(theta, out_size):
    th, tw = out_size
    B = theta.shape[0]
    ys, xs = np.meshgrid(np.linspace(-1, 1, th, dtype=theta.dtype),
                         np.linspace(-1, 1, tw, dtype=theta.dtype), indexing='ij')
    coords = np.concatenate([xs[None], ys[None], np.ones((1, th, tw), dtype=theta.dtype)], axis=0)
    grid = theta.dot(coords.reshape(3, th * tw)).reshape(B, 2, th, tw)
    return grid.astype(theta.dtype, copy=False), coords


def st_grid_bwd(coords, ggrid):
    B, _, th, tw = ggrid.shape
    gtheta = ggrid.reshape(B, 2, th * tw).dot(coords.reshape(3, th * tw).T)
    return gtheta.astype(ggrid.dtype, copy=False)


def _st_sampler_setup(x, grid):
    B, C, H, W = x.shape
    g = grid.reshape(grid.shape[:2] + (-1,))
    u = g[:, 0]
    v = g[:, 1]
    x_pad = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)), mode='constant')
    u = (u + 1) * (W - 1) / 2 + 1
    v = (v + 1) * (H - 1) / 2 + 1
    u_clipped = u.clip(0, W + 1)
    v_clipped = v.clip(0, H + 1)
    u0 = np.floor(u_clipped).astype(np.int32).clip(0, W)
    u1 = u0 + 1
    v0 = np.floor(v_clipped).astype(np.int32).clip(0, H)
    v1 = v0 + 1
    return x_pad, u, v, u_clipped, v_clipped, u0, u1, v0, v1


def _st_gather(x_pad, vi, ui):
    B = x_pad.shape[0]
    return np.concatenate([np.expand_dims(x_pad[b, :, vi[b], ui[b]], axis=0) for b in range(B)], axis=0)


def st_sampler_fwd(x, grid):
    B, C, H, W = x.shape
    _, _, oh, ow = grid.shape
    x_pad, u, v, uc, vc, u0, u1, v0, v1 = _st_sampler_setup(x, grid)
    dt = x.dtype
    w1 = ((u1 - uc) * (v1 - vc)).astype(dt)
    w2 = ((uc - u0) * (v1 - vc)).astype(dt)
    w3 = ((u1 - uc) * (vc - v0)).astype(dt)
    w4 = ((uc - u0) * (vc - v0)).astype(dt)
    y = w1[:, :, None] * _st_gather(x_pad, v0, u0)
    y += w2[:, :, None] * _st_gather(x_pad, v0, u1)
    y += w3[:, :, None] * _st_gather(x_pad, v1, u0)
    y += w4[:, :, None] * _st_gather(x_pad, v1, u1)
    return y.reshape(B, oh, ow, C).transpose(0, 3, 1, 2)


def st_sampler_bwd_grid(x, grid, gy):
    """Gradient w.r.t. the sampling grid (the image is a leaf on this path)."""
    B, C, H, W = x.shape
    _, _, oh, ow = grid.shape
    x_pad, u, v, uc, vc, u0, u1, v0, v1 = _st_sampler_setup(x, grid)
    dt = gy.dtype
    wu0 = (uc - u0).astype(dt)
    wu1 = (u1 - uc).astype(dt)
    wv0 = (vc - v0).astype(dt)
    wv1 = (v1 - vc).astype(dt)
    x1 = _st_gather(x_pad, v0, u0)
    x2 = _st_gather(x_pad, v0, u1)
    x3 = _st_gather(x_pad, v1, u0)
    x4 = _st_gather(x_pad, v1, u1)
    gu = -wv1[:, :, None] * x1
    gu += wv1[:, :, None] * x2
    gu -= wv0[:, :, None] * x3
    gu += wv0[:, :, None] * x4
    gv = -wu1[:, :, None] * x1
    gv -= wu0[:, :, None] * x2
    gv += wu1[:, :, None] * x3
    gv += wu0[:, :, None] * x4
    gu = gu.reshape(B, oh, ow, C).transpose(0, 3, 1, 2)
    gv = gv.reshape(B, oh, ow, C).transpose(0, 3, 1, 2)
    gu = (gu * gy).sum(axis=1)
    gv = (gv * gy).sum(axis=1)
    ur = u.reshape(gu.shape)
    vr = v.reshape(gv.shape)
    gu = gu / 2. * (W - 1) * (ur > 0) * (ur < (W + 1))
    gv = gv / 2. * (H - 1) * (vr > 0) * (vr < (H + 1))
    return np.concatenate((gu[:, None], gv[:, None]), axis=1).astype(dt, copy=False)


# --------------------------------------------------------------------------- #
# sigmoid / MSE (common/net.py:90 ; sheep_updater.py:43,60)
# --------------------------------------------------------------------------- #
def sigmoid(x):
    half = x.dtype.type(0.5)
    return np.tanh(x * half) * half + half


def sigmoid_bwd(y, gy):
    return gy * y * (1 - y)


def mse_fwd(x0, x1):
    diff = (x0 - x1).ravel()
    return x0.dtype.type(diff.dot(diff) / diff.size)


def mse_bwd(x0, x1, gloss=1.0):
    diff = x0 - x1
    return (gloss * 2.0 / diff.size) * diff


# --------------------------------------------------------------------------- #
# grid regularisers: common/utils.py:142-178 (direction), :301-316 (out of image)
# --------------------------------------------------------------------------- #
def direction_loss(grids, image_size):
    """Returns (loss, ggrids).  image_size = (height, width)."""
    H, W = image_size
    B, _, th, tw = grids.shape
    g = (grids + 1) / 2
    xp_ = g[:, 0] * W
    yp_ = g[:, 1] * H
    tlx, trx = xp_[:, 0, 0], xp_[:, 0, tw - 1]
    tly, bly = yp_[:, 0, 0], yp_[:, th - 1, 0]
    d1 = tly - bly
    d2 = tlx - trx
    loss = np.maximum(d1, 0).mean() + np.maximum(d2, 0).mean()
    gg = np.zeros_like(grids)
    # F.maximum(distance, zeros) (common/utils.py:169,175): Chainer 4.1's Maximum routes the gradient with `x1 >= x2`
    # (gx1 = gy * (x1 >= x2), gx2 = gy * (x1 < x2)) -- at an exact tie it goes to the FIRST argument, here `distance`
    m1 = (d1 >= 0).astype(grids.dtype) / B * (H / 2.0)
    m2 = (d2 >= 0).astype(grids.dtype) / B * (W / 2.0)
    gg[:, 1, 0, 0] += m1
    gg[:, 1, th - 1, 0] -= m1
    gg[:, 0, 0, 0] += m2
    gg[:, 0, 0, tw - 1] -= m2
    return grids.dtype.type(loss), gg


def out_of_image_loss(grids):
    B, _, th, tw = grids.shape
    xs, ys = grids[:, 0], grids[:, 1]
    tlx, tly, trx, bly = xs[:, 0, 0], ys[:, 0, 0], xs[:, 0, tw - 1], ys[:, th - 1, 0]
    bbox = np.concatenate([tlx, tly, trx, bly], axis=0)
    top = bbox + 1
    bottom = bbox - 1
    loss = np.abs(np.minimum(top, 0)).sum() + np.maximum(bottom, 0).sum()
    # common/utils.py:312-313.  bottom: F.maximum(bottom_loss, zeros) passes the gradient where bottom_loss >= 0 (tie included,
    # see direction_loss).  top: F.absolute(F.minimum(top_loss, zeros)) -- Minimum passes it where top_loss <= 0, but
    # Absolute's backward is sign(x) * gy and sign(0) = 0: at the tie top_loss == 0 nothing comes back.
    gb = -(top < 0).astype(grids.dtype) + (bottom >= 0).astype(grids.dtype)
    gg = np.zeros_like(grids)
    gg[:, 0, 0, 0] += gb[0:B]
    gg[:, 1, 0, 0] += gb[B:2 * B]
    gg[:, 0, 0, tw - 1] += gb[2 * B:3 * B]
    gg[:, 1, th - 1, 0] += gb[3 * B:4 * B]
    return grids.dtype.type(loss), gg


# --------------------------------------------------------------------------- #
# Adam with AMSGrad, Chainer 4.1.0 placement of eps / bias correction
#   (train_sheep_localizer.py:130-134 ; sheep_updater.py:52,66)
# --------------------------------------------------------------------------- #
def adam_lr(alpha, beta1, beta2, t):
    fix1 = 1. - math.pow(beta1, t)
    fix2 = 1. - math.pow(beta2, t)
    return alpha * math.sqrt(fix2) / fix1


def adam_amsgrad_update(p, g, m, v, vhat, t, alpha=0.001, beta1=0.9, beta2=0.999,
                        eps=1e-8, eta=1.0, weight_decay_rate=0.0, amsgrad=True):
    """In-place update of (p, m, v, vhat); ``t`` is the 1-based step count.  ``amsgrad=False`` (Chainer's default; the
    trainer passes True, train_sheep_localizer.py:130-134): the denominator is sqrt(v), ``vhat`` is left alone."""
    dt = p.dtype.type
    m += dt(1 - beta1) * (g - m)
    v += dt(1 - beta2) * (g * g - v)
    if amsgrad:
        np.maximum(vhat, v, out=vhat)
    else:
        vhat = v
    lr = adam_lr(alpha, beta1, beta2, t)
    p -= dt(eta) * (dt(lr) * m / (np.sqrt(vhat) + dt(eps)) + dt(weight_decay_rate) * p)
