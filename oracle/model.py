"""CPU oracle: the LoANs localizer / assessor graphs and the joint update step.

TEST INFRASTRUCTURE ONLY (see chainer_ops.py header).  PARITY UNPINNED: no
reference test pins these results; see DESIGN.md §Oracle.

Topology / wiring follows the reference files (hyper-parameters and order of
operations), the arithmetic follows chainer_ops.py:

* ``ResNet(18)`` variant ............ sheep/resnet.py:6-160
* ``SheepLocalizer`` ................ sheep/sheep_localizer.py:18-117
* ``ResnetAssessor`` ................ common/net.py:6-90
* ``SheepAssessor.update_core`` ..... sheep/sheep_updater.py:26-68
* optimiser wiring .................. train_sheep_localizer.py:130-134

Parameters live in plain dicts keyed by the Chainer npz paths
(``feature_extractor/res2/0/conv1/W`` ...), weights in OIHW.
"""
import numpy as np

from . import chainer_ops as C

STAGES = (('res2', 64, 1), ('res3', 128, 2), ('res4', 256, 2), ('res5', 512, 2))


# --------------------------------------------------------------------------- #
# parameter construction (initialisers: HeNormal sheep/resnet.py:10,
# Normal(0.02) common/net.py:72, param_predictor sheep_localizer.py:28-33)
# --------------------------------------------------------------------------- #
def _he(rng, shape):
    fan_in = int(np.prod(shape[1:]))
    return (rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)


def _bn(p, prefix, ch):
    p[prefix + '/gamma'] = np.ones(ch, np.float32)
    p[prefix + '/beta'] = np.zeros(ch, np.float32)
    p[prefix + '/avg_mean'] = np.zeros(ch, np.float32)
    p[prefix + '/avg_var'] = np.ones(ch, np.float32)
    p[prefix + '/N'] = np.array(0)


def _basic_block(p, rng, prefix, cin, ch):
    a = prefix + '/0'
    p[a + '/conv1/W'] = _he(rng, (ch, cin, 3, 3)); _bn(p, a + '/bn1', ch)
    p[a + '/conv2/W'] = _he(rng, (ch, ch, 3, 3)); _bn(p, a + '/bn2', ch)
    p[a + '/conv3/W'] = _he(rng, (ch, cin, 3, 3)); _bn(p, a + '/bn3', ch)
    b = prefix + '/1'
    p[b + '/conv1/W'] = _he(rng, (ch, ch, 3, 3)); _bn(p, b + '/bn1', ch)
    p[b + '/conv2/W'] = _he(rng, (ch, ch, 3, 3)); _bn(p, b + '/bn2', ch)


def init_localizer_params(rng, with_res67=True, predictor_w_std=0.0):
    """Fresh ``SheepLocalizer`` state.  ``predictor_w_std`` > 0 gives the
    non-degenerate seeded ``param_predictor.W`` the parity runs need
    (SURVEY §8d); 0 reproduces the reference's zero init."""
    p = {}
    fe = 'feature_extractor'
    p[fe + '/conv1/W'] = _he(rng, (64, 3, 7, 7))
    p[fe + '/conv1/b'] = np.zeros(64, np.float32)
    _bn(p, fe + '/bn1', 64)
    cin = 64
    for name, ch, _ in STAGES:
        _basic_block(p, rng, fe + '/' + name, cin, ch)
        cin = ch
    if with_res67:
        _basic_block(p, rng, 'res6', 512, 512)
        _basic_block(p, rng, 'res7', 512, 512)
    W = np.zeros((6, 512), np.float32)
    if predictor_w_std > 0:
        W = (rng.standard_normal((6, 512)) * predictor_w_std).astype(np.float32)
    p['param_predictor/W'] = W
    p['param_predictor/b'] = np.array([0.8, 0, 0, 0, 0.8, 0], np.float32)
    return p


def init_assessor_params(rng, target_size=(75, 75), ch=128, wscale=0.02):
    def n(shape):
        return (rng.standard_normal(shape) * wscale).astype(np.float32)
    p = {}
    p['r0/c0/W'] = n((ch, 3, 3, 3)); p['r0/c1/W'] = n((ch, ch, 4, 4)); p['r0/cs/W'] = n((ch, 3, 4, 4))
    p['r1/c0/W'] = n((ch, ch, 3, 3)); p['r1/c1/W'] = n((ch, ch, 4, 4)); p['r1/cs/W'] = n((ch, ch, 4, 4))
    for r in ('r2', 'r3'):
        p[r + '/c0/W'] = n((ch, ch, 3, 3)); p[r + '/c1/W'] = n((ch, ch, 3, 3))
    h = C.conv_outsize(C.conv_outsize(target_size[0], 4, 2, 1), 4, 2, 1)
    w = C.conv_outsize(C.conv_outsize(target_size[1], 4, 2, 1), 4, 2, 1)
    p['l4/W'] = n((1, ch * h * w))
    return p


def is_trainable(key):
    return not key.endswith(('/avg_mean', '/avg_var', '/N'))


def cast_params(p, dtype):
    return {k: (v.astype(dtype) if v.dtype.kind == 'f' else v.copy()) for k, v in p.items()}


# --------------------------------------------------------------------------- #
# bf16-storage emulation (BASELINE configs 3 / 5): the HIP bf16 arm keeps every activation / gradient tensor of the
# residual stages and of the assessor in bf16.  Inside `emulate_bf16_storage()` the oracle rounds its
# tensors to bf16 at exactly the places where that arm STORES one -- conv outputs (statistics still from the unrounded
# accumulators), the fused BN-apply / ReLU / residual-sum outputs, the fused dgrad (+ mask / addend) outputs, BN-backward
# outputs, the bf16 operand copies of the weights and of the preprocessed frames -- while every reduction, coefficient,
# parameter gradient and the optimiser stay in the oracle's own precision.  `_q` is the identity otherwise.
# --------------------------------------------------------------------------- #
import contextlib

_Q = None


def _q(a):
    return a if _Q is None else _Q(a)


@contextlib.contextmanager
def emulate_bf16_storage():
    global _Q
    old, _Q = _Q, C.round_bf16
    try:
        yield
    finally:
        _Q = old


# --------------------------------------------------------------------------- #
# building blocks with explicit forward / backward
# --------------------------------------------------------------------------- #
class _ConvBN:
    """conv (optionally biased) followed by BatchNormalization."""

    def __init__(self, p, conv, bn, stride, pad, train):
        self.p, self.conv, self.bn, self.stride, self.pad, self.train = p, conv, bn, stride, pad, train

    def fwd(self, x):
        p = self.p
        self.x_shape = x.shape
        self.has_bias = (self.conv + '/b') in p
        c, self.col = C.conv2d_fwd(x, _q(p[self.conv + '/W']), p.get(self.conv + '/b'), self.stride, self.pad)
        if self.train:
            y, self.ctx = C.bn_fwd_train(c, p[self.bn + '/gamma'], p[self.bn + '/beta'],
                                         p[self.bn + '/avg_mean'], p[self.bn + '/avg_var'],
                                         x_apply=None if _Q is None else _q(c))
        else:
            y = C.bn_fwd_test(_q(c), p[self.bn + '/gamma'], p[self.bn + '/beta'],
                              p[self.bn + '/avg_mean'], p[self.bn + '/avg_var'])
        return y

    def bwd(self, gy, grads, need_gx=True):
        p = self.p
        if self.train:
            gc, gg, gb = C.bn_bwd(self.ctx, p[self.bn + '/gamma'], gy)
            _acc(grads, self.bn + '/gamma', gg)
            _acc(grads, self.bn + '/beta', gb)
        else:
            gc = C.bn_bwd_test(p[self.bn + '/gamma'], p[self.bn + '/avg_var'], gy)
        gc = _q(gc)
        gx, gW, gbias = C.conv2d_bwd(self.x_shape, self.col, _q(p[self.conv + '/W']), gc,
                                     self.stride, self.pad, self.has_bias, need_gx)
        _acc(grads, self.conv + '/W', gW)
        if self.has_bias:
            _acc(grads, self.conv + '/b', gbias)
        return gx


def _acc(grads, key, g):
    if grads is None:
        return
    if key in grads:
        grads[key] = grads[key] + g
    else:
        grads[key] = g


class _BasicA:  # sheep/resnet.py:121-141
    def __init__(self, p, prefix, stride, train):
        self.c1 = _ConvBN(p, prefix + '/conv1', prefix + '/bn1', stride, 1, train)
        self.c2 = _ConvBN(p, prefix + '/conv2', prefix + '/bn2', 1, 1, train)
        self.c3 = _ConvBN(p, prefix + '/conv3', prefix + '/bn3', stride, 1, train)

    def fwd(self, x):
        self.h1 = _q(C.relu(self.c1.fwd(x)))
        a = self.c2.fwd(self.h1)
        b = self.c3.fwd(x)
        self.out = _q(C.relu(a + b))
        return self.out

    def bwd(self, gy, grads):
        gz = gy * (self.out > 0)
        gx = _q(self.c3.bwd(gz, grads))
        gh1 = _q(self.c2.bwd(gz, grads))
        gx = _q(gx + self.c1.bwd(gh1 * (self.h1 > 0), grads))
        return gx


class _BasicB:  # sheep/resnet.py:144-160
    def __init__(self, p, prefix, train):
        self.c1 = _ConvBN(p, prefix + '/conv1', prefix + '/bn1', 1, 1, train)
        self.c2 = _ConvBN(p, prefix + '/conv2', prefix + '/bn2', 1, 1, train)

    def fwd(self, x):
        self.h1 = _q(C.relu(self.c1.fwd(x)))
        self.out = _q(C.relu(self.c2.fwd(self.h1) + x))
        return self.out

    def bwd(self, gy, grads):
        gz = gy * (self.out > 0)
        gh1 = _q(self.c2.bwd(gz, grads))
        return _q(gz + self.c1.bwd(gh1 * (self.h1 > 0), grads))


class _ResUnit:
    """relu(bn_n(conv_n(... relu(bn_1(conv_1(x))))) + shortcut(x)); stages / shortcut = (conv_key, bn_key, stride, pad).
    Bottleneck units of Chainer's ResNet50Layers (mirrored by sheep/resnet.py:163-216) and chainercv's ResBlock."""

    def __init__(self, p, stages, shortcut, train):
        self.stages = [_ConvBN(p, c, b, s, pad, train) for c, b, s, pad in stages]
        self.shortcut = None if shortcut is None else _ConvBN(p, shortcut[0], shortcut[1], shortcut[2], shortcut[3], train)

    def fwd(self, x):
        self.hs = []
        h = x
        for i, st in enumerate(self.stages):
            h = st.fwd(h)
            if i < len(self.stages) - 1:
                h = _q(C.relu(h))
                self.hs.append(h)
        sc = self.shortcut.fwd(x) if self.shortcut is not None else x
        self.out = _q(C.relu(h + sc))
        return self.out

    def bwd(self, gy, grads):
        gz = gy * (self.out > 0)
        gx = _q(self.shortcut.bwd(gz, grads)) if self.shortcut is not None else gz
        g = gz
        for i in range(len(self.stages) - 1, 0, -1):
            g = _q(self.stages[i].bwd(g, grads)) * (self.hs[i - 1] > 0)
        return _q(gx + self.stages[0].bwd(g, grads))


RESNET50_STAGES = (('res2', 3, 64, 64, 256, 1), ('res3', 4, 256, 128, 512, 2),
                   ('res4', 6, 512, 256, 1024, 2), ('res5', 3, 1024, 512, 2048, 2))


def init_resnet50_localizer_params(rng, predictor_w_std=0.0):
    """``Resnet50SheepLocalizer`` state (sheep_localizer.py:120-141): HeNormal(1.0) backbone (the Caffe
    weights of pretrained_model='auto' are not available offline), HeNormal(fan_out) res6 / res7."""
    p = {}
    fe = 'feature_extractor'
    p[fe + '/conv1/W'] = _he(rng, (64, 3, 7, 7))
    p[fe + '/conv1/b'] = np.zeros(64, np.float32)
    _bn(p, fe + '/bn1', 64)
    for name, n, cin, mid, cout, _ in RESNET50_STAGES:
        a = '%s/%s/a' % (fe, name)
        p[a + '/conv1/W'] = _he(rng, (mid, cin, 1, 1)); _bn(p, a + '/bn1', mid)
        p[a + '/conv2/W'] = _he(rng, (mid, mid, 3, 3)); _bn(p, a + '/bn2', mid)
        p[a + '/conv3/W'] = _he(rng, (cout, mid, 1, 1)); _bn(p, a + '/bn3', cout)
        p[a + '/conv4/W'] = _he(rng, (cout, cin, 1, 1)); _bn(p, a + '/bn4', cout)
        for i in range(1, n):
            b = '%s/%s/b%d' % (fe, name, i)
            p[b + '/conv1/W'] = _he(rng, (mid, cout, 1, 1)); _bn(p, b + '/bn1', mid)
            p[b + '/conv2/W'] = _he(rng, (mid, mid, 3, 3)); _bn(p, b + '/bn2', mid)
            p[b + '/conv3/W'] = _he(rng, (cout, mid, 1, 1)); _bn(p, b + '/bn3', cout)

    def he_out(shape):
        fan_out = shape[0] * int(np.prod(shape[2:]))
        return (rng.standard_normal(shape) * np.sqrt(2.0 / fan_out)).astype(np.float32)
    for name in ('res6', 'res7'):
        for blk, cin in (('a', 2048), ('b1', 2048)):
            q = '%s/%s' % (name, blk)
            p[q + '/conv1/conv/W'] = he_out((1024, cin, 1, 1)); _bn(p, q + '/conv1/bn', 1024)
            p[q + '/conv2/conv/W'] = he_out((1024, 1024, 3, 3)); _bn(p, q + '/conv2/bn', 1024)
            p[q + '/conv3/conv/W'] = he_out((2048, 1024, 1, 1)); _bn(p, q + '/conv3/bn', 2048)
            if blk == 'a':
                p[q + '/residual_conv/conv/W'] = he_out((2048, cin, 1, 1)); _bn(p, q + '/residual_conv/bn', 2048)
    W = np.zeros((6, 2048), np.float32)
    if predictor_w_std > 0:
        W = (rng.standard_normal((6, 2048)) * predictor_w_std).astype(np.float32)
    p['param_predictor/W'] = W
    p['param_predictor/b'] = np.array([0.8, 0, 0, 0, 0.8, 0], np.float32)
    return p


class Localizer:
    """``SheepLocalizer.__call__`` with an explicit backward (sheep_localizer.py:41-70)."""

    def __init__(self, params, out_size, train=True, rng=None):
        self.p, self.out_size, self.train, self.rng = params, tuple(out_size), train, rng

    def forward(self, images):
        p, train = self.p, self.train
        self.images = images
        H = images.shape[-2]
        x = _q(C.prepare_images(images))
        self.blocks = []
        fe = 'feature_extractor'
        self.stem = _ConvBN(p, fe + '/conv1', fe + '/bn1', 2, 3, train)
        self.stem_relu = _q(C.relu(self.stem.fwd(x)))
        h, self.pool_idx = C.max_pool_fwd(self.stem_relu, 3, 2, 0)
        for blk in self._make_blocks(H):
            h = blk.fwd(h)
            self.blocks.append(blk)
        self.feat = h
        self.pooled = C.gap_fwd(h)
        theta = C.linear_fwd(self.pooled, p['param_predictor/W'], p['param_predictor/b']).reshape(-1, 2, 3)
        self.mask = C.rotation_dropout_mask(theta, 0.0, train, self.rng)   # sheep_localizer.py:61
        self.theta = theta * self.mask
        self.points, self.coords = C.st_grid_fwd(self.theta, self.out_size)
        self.rois = C.st_sampler_fwd(images, self.points)
        return self.rois, self.points

    def _make_blocks(self, H):
        p, train, fe = self.p, self.train, 'feature_extractor'
        names = [(fe + '/' + n, s) for n, _, s in STAGES]
        if H > 224:
            names.append(('res6', 2))
            if H > 300:
                names.append(('res7', 2))
        out = []
        for prefix, stride in names:
            out += [_BasicA(p, prefix + '/0', stride, train), _BasicB(p, prefix + '/1', train)]
        return out

    def backward(self, g_rois, g_points, grads):
        p = self.p
        gpoints = np.zeros_like(self.points)
        if g_points is not None:
            gpoints = gpoints + g_points
        if g_rois is not None:
            gpoints = gpoints + C.st_sampler_bwd_grid(self.images, self.points, g_rois)
        gtheta = C.st_grid_bwd(self.coords, gpoints) * self.mask
        gpooled, gW, gb = C.linear_bwd(self.pooled, p['param_predictor/W'], gtheta.reshape(-1, 6), True)
        _acc(grads, 'param_predictor/W', gW)
        _acc(grads, 'param_predictor/b', gb)
        g = _q(C.gap_bwd(self.feat.shape, gpooled))
        for blk in reversed(self.blocks):
            g = blk.bwd(g, grads)
        g = C.max_pool_bwd(self.stem_relu.shape, self.pool_idx, g, 3, 2, 0)
        g = g * (self.stem_relu > 0)
        self.stem.bwd(g, grads, need_gx=False)

    # sheep_localizer.py:84-97
    def corners_px(self, points, image_hw):
        H, W = image_hw
        top, left = points[:, 1, 0, 0], points[:, 0, 0, 0]
        bottom, right = points[:, 1, -1, -1], points[:, 0, -1, -1]
        bb = (np.stack([top, left, bottom, right], axis=1) + 1) / 2
        bb[:, ::2] *= H
        bb[:, 1::2] *= W
        return bb


class Localizer50(Localizer):
    """``Resnet50SheepLocalizer.__call__`` (sheep_localizer.py:143-178): bottleneck backbone, stride on the
    first 1x1 conv; res6 / res7 are chainercv ResBlocks with the stride on the 3x3 conv."""

    def _make_blocks(self, H):
        p, train, fe = self.p, self.train, 'feature_extractor'
        out = []
        for name, n, cin, mid, cout, stride in RESNET50_STAGES:
            a = '%s/%s/a' % (fe, name)
            out.append(_ResUnit(p, [(a + '/conv1', a + '/bn1', stride, 0), (a + '/conv2', a + '/bn2', 1, 1),
                                    (a + '/conv3', a + '/bn3', 1, 0)], (a + '/conv4', a + '/bn4', stride, 0), train))
            for i in range(1, n):
                b = '%s/%s/b%d' % (fe, name, i)
                out.append(_ResUnit(p, [(b + '/conv1', b + '/bn1', 1, 0), (b + '/conv2', b + '/bn2', 1, 1),
                                        (b + '/conv3', b + '/bn3', 1, 0)], None, train))
        extra = (['res6'] if H > 224 else []) + (['res7'] if H > 300 else [])
        for name in extra:
            for blk in ('a', 'b1'):
                q = '%s/%s' % (name, blk)
                s = 2 if blk == 'a' else 1
                sc = (q + '/residual_conv/conv', q + '/residual_conv/bn', 2, 0) if blk == 'a' else None
                out.append(_ResUnit(p, [(q + '/conv1/conv', q + '/conv1/bn', 1, 0), (q + '/conv2/conv', q + '/conv2/bn', s, 1),
                                        (q + '/conv3/conv', q + '/conv3/bn', 1, 0)], sc, train))
        return out


class Assessor:
    """``ResnetAssessor.__call__`` with explicit backward (common/net.py:83-90)."""

    def __init__(self, params):
        self.p = params

    def _conv(self, key, x, stride, pad):
        y, col = C.conv2d_fwd(x, _q(self.p[key + '/W']), None, stride, pad)
        self._ctx[key] = (x.shape, col, stride, pad)
        return y

    def _conv_bwd(self, key, gy, grads, need_gx=True, round_w=True):
        shape, col, stride, pad = self._ctx[key]
        W = self.p[key + '/W']
        gx, gW, _ = C.conv2d_bwd(shape, col, _q(W) if round_w else W, gy, stride, pad, False, need_gx)
        _acc(grads, key + '/W', gW)
        return gx

    def forward(self, x):
        self._ctx = {}
        x = _q(x)            # bf16 emulation: the crops stay fp32 in HBM, the first block rounds them while it stages them
        self.x = x
        # r0 = DownResBlock1 (net.py:19-25)
        self.r0_h1 = _q(self._conv('r0/c0', x, 1, 1))
        h = _q(self._conv('r0/c1', C.relu(self.r0_h1), 2, 1) + _q(self._conv('r0/cs', x, 2, 1)))
        self.h1 = h
        # r1 = DownResBlock2 (net.py:41-47)
        self.r1_h1 = _q(self._conv('r1/c0', C.relu(h), 1, 1))
        h = _q(self._conv('r1/c1', C.relu(self.r1_h1), 2, 1) + _q(self._conv('r1/cs', h, 2, 1)))
        self.h2 = h
        # r2, r3 = DownResBlock3 (net.py:62-67)
        self.r2_h1 = _q(self._conv('r2/c0', C.relu(h), 1, 1))
        h = _q(self._conv('r2/c1', C.relu(self.r2_h1), 1, 1) + h)
        self.h3 = h
        self.r3_h1 = _q(self._conv('r3/c0', C.relu(h), 1, 1))
        h = _q(self._conv('r3/c1', C.relu(self.r3_h1), 1, 1) + h)
        self.h4 = h
        self.hr = C.relu(h)
        self.y = C.sigmoid(C.linear_fwd(self.hr, self.p['l4/W'], None))
        return self.y

    def backward(self, gy, grads, need_gx=True):
        """``grads=None`` skips accumulating parameter gradients (chain A of
        update_core: the assessor's wgrads are computed by Chainer but cleared
        before use, sheep_updater.py:48-51,63)."""
        gz = C.sigmoid_bwd(self.y, gy)
        ghr, gW, _ = C.linear_bwd(self.hr, self.p['l4/W'], gz, False)
        _acc(grads, 'l4/W', gW)
        g = _q(ghr * (self.h4 > 0))
        # r3
        g1 = _q(self._conv_bwd('r3/c1', g, grads) * (self.r3_h1 > 0))
        g = _q(g + self._conv_bwd('r3/c0', g1, grads) * (self.h3 > 0))
        # r2
        g1 = _q(self._conv_bwd('r2/c1', g, grads) * (self.r2_h1 > 0))
        g = _q(g + self._conv_bwd('r2/c0', g1, grads) * (self.h2 > 0))
        # r1
        g1 = _q(self._conv_bwd('r1/c1', g, grads) * (self.r1_h1 > 0))
        g = _q(_q(self._conv_bwd('r1/cs', g, grads)) + self._conv_bwd('r1/c0', g1, grads) * (self.h1 > 0))
        # r0
        g1 = _q(self._conv_bwd('r0/c1', g, grads) * (self.r0_h1 > 0))
        if not need_gx:
            self._conv_bwd('r0/cs', g, grads, need_gx=False)
            self._conv_bwd('r0/c0', g1, grads, need_gx=False)
            return None
        # the gradient w.r.t. the 4-channel crops: fp32 out; in the bf16 emulation the weights are the bf16-rounded ones like
        # everywhere else (round 6: loans_crop_dgrad_bf16_f32 contracts on bf16 MFMAs; until then it read the fp32 masters)
        return self._conv_bwd('r0/cs', g, grads) + self._conv_bwd('r0/c0', g1, grads)


# --------------------------------------------------------------------------- #
# optimiser + the joint step
# --------------------------------------------------------------------------- #
class AdamAMSGrad:
    """chainer.optimizers.Adam(alpha, amsgrad=True) over a parameter dict."""

    def __init__(self, params, alpha=0.001):
        self.params, self.alpha, self.t = params, alpha, 0
        self.state = {k: tuple(np.zeros_like(v) for _ in range(3))
                      for k, v in params.items() if is_trainable(k)}

    def update(self, grads):
        self.t += 1
        for k, (m, v, vhat) in self.state.items():
            g = grads.get(k)
            if g is None:                  # reallocate_cleared_grads -> zeros
                g = np.zeros_like(self.params[k])
            C.adam_amsgrad_update(self.params[k], g.astype(self.params[k].dtype, copy=False),
                                  m, v, vhat, self.t, alpha=self.alpha)


def update_core(loc_params, dis_params, opt_gen, opt_dis, fake_images, real_images, labels,
                out_size, localizer_target=1.0, freeze_discriminator=False, rng=None,
                return_grads=False, localizer_cls=None, oob_scale=1.0):
    """One ``SheepAssessor.update_core`` (sheep_updater.py:26-68).  Mutates the
    parameter dicts / optimiser states in place; returns the reported losses and
    the tensors the parity tests compare."""
    dtype = fake_images.dtype
    dis = Assessor(dis_params)
    y_real = dis.forward(real_images)                                   # :35
    dis_real = dis

    loc = (localizer_cls or Localizer)(loc_params, out_size, train=True, rng=rng)
    x_fake, bboxes = loc.forward(fake_images)                           # :39
    dis_fake = Assessor(dis_params)
    y_fake = dis_fake.forward(x_fake)                                   # :40

    target = np.full((len(y_fake), 1), localizer_target, dtype=dtype)   # :42
    loss_localizer = C.mse_fwd(y_fake, target)                          # :43
    H, W = fake_images.shape[-2:]
    l_dir, g_dir = C.direction_loss(bboxes, (H, W))                     # :45-46
    l_out, g_out = C.out_of_image_loss(bboxes)
    # oob_scale != 1 only in the data-parallel tests: OutOfImageLoss is a SUM over the batch (common/utils.py:315), a
    # rank of a k-way data-parallel run scales it by k so that the averaged gradients equal the global-batch gradient
    l_out, g_out = l_out * oob_scale, g_out * oob_scale
    loss_localizer = loss_localizer + l_dir + l_out

    loc_grads = {}
    g_yfake = C.mse_bwd(y_fake, target)
    g_xfake = dis_fake.backward(g_yfake, None, need_gx=True)            # :48-51
    loc.backward(g_xfake, g_dir + g_out, loc_grads)
    opt_gen.update(loc_grads)                                           # :52

    loss_dis = C.mse_fwd(y_real, labels)                                # :60
    dis_grads = {}
    if not freeze_discriminator:                                        # :62-66
        dis_real.backward(C.mse_bwd(y_real, labels), dis_grads, need_gx=False)
        opt_dis.update(dis_grads)

    out = dict(loss_localizer=float(loss_localizer), loss_dis=float(loss_dis),
               theta=loc.theta, points=bboxes, rois=x_fake, y_fake=y_fake, y_real=y_real)
    if return_grads:
        out['loc_grads'] = loc_grads
        out['dis_grads'] = dis_grads
    return out
