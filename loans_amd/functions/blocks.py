"""Fused residual-block functions: each is ONE autograd node whose forward and
backward are a short, fixed list of kernel launches (conv + BN statistics in
the conv epilogue, BN-apply + ReLU + residual sum in one streaming pass,
ReLU-backward masks and shortcut sums folded into the dgrad epilogues).

Topology follows the reference: stem sheep/resnet.py:72-73, BasicA :136-141,
BasicB :156-160, DownResBlock1/2/3 common/net.py:19-25,41-47,62-67.
Activations are NHWC; parameter gradients are accumulated straight into the
model's gradient arena (``Parameter.grad_view``)."""
import torch

from .. import ops
from ..runtime.core import Function, config


def _zeros_stats(C, device):
    return ops.stats_buffer(C, device)


def _wgrad_enabled(param):
    """Chainer computes the assessor's weight gradients in the localizer's backward pass
    and clears them unused (sheep_updater.py:48-51,63); with updates disabled we skip
    that dead work."""
    return param.update_rule.enabled


class _ConvBN:
    """conv -> BN statistics (train) or running-stat coefficients (test)."""

    @staticmethod
    def forward(x, conv, bn, W, b, gamma, beta):
        # the stem (fp32 frames) is where the bf16 region of ops.STORAGE == 'bf16' begins: its conv writes bf16
        o16 = ops.STORAGE == 'bf16' and x.dtype == torch.float32
        B, H, Wd, _ = x.shape
        if conv.dense_rows:
            if not hasattr(x, 'frame_hw'):
                raise ValueError('a dense-row stem takes the padded frame buffer of ops.prep_images(images, geometry)')
            H, Wd = x.frame_hw
        geo = conv.geometry(B, H, Wd)
        if config.train:
            stats = _zeros_stats(conv.out_channels, x.device)
            c = ops.conv_fprop(x, W, geo, bias=b, stats=stats, out_bf16=o16)
            st = ops.bn_finalize(stats, B * geo.Ho * geo.Wo, gamma, beta, bn.avg_mean, bn.avg_var)
        else:
            c = ops.conv_fprop(x, W, geo, bias=b, out_bf16=o16)
            st = ops.bn_eval_coeffs(gamma, beta, bn.avg_mean, bn.avg_var)
        return c, st, geo


def _require_train():
    if not config.train:
        raise RuntimeError('backward through test-mode BatchNormalization is not on the training path')


class StageBoundary(Function):
    """Identity between two stages of a backbone that tells a data-parallel optimiser when a stage's parameter gradients are
    complete: its backward runs once every function ABOVE it has run its own, i.e. when the weight gradients of the stages
    behind it are enqueued -- the optimiser then starts the all-reduce of that part of the gradient arena while the backward of
    the stages in front of it is still being issued (loans_amd/parallel.py, SURVEY 8e: "bucket + overlap with backward").
    Only inserted while a gradient exchange is active; the values that flow through are untouched."""

    def __init__(self, hook, name):
        self.hook, self.name = hook, name

    def forward(self, inputs):
        return inputs[0]

    def backward(self, inputs, grad_outputs):
        self.hook(self.name)
        return grad_outputs[0]


def stage_boundary(link, name, h):
    """``h`` unchanged, with a StageBoundary in front of stage ``name`` when ``link`` has an exchange hook attached"""
    hook = link.__dict__.get('_stage_hook')
    if hook is None or not config.enable_backprop or not config.train:
        return h
    return StageBoundary(hook, name)(h)


class StemFunction(Function):
    """conv1(7x7/2, bias) -> bn1 -> relu -> max_pool(3, 2, cover_all)   (sheep/resnet.py:72-73).
    inputs: x (preprocessed frames: the zero-padded packed-RGB buffer of ops.prep_images(images, geometry) for a
    dense-row conv1, NHWC4 otherwise; no gradient), W, b, gamma, beta."""

    def __init__(self, conv, bn):
        self.conv, self.bn = conv, bn

    def forward(self, inputs):
        x, W, b, gamma, beta = inputs
        self.x = x
        self.c, self.st, self.geo = _ConvBN.forward(x, self.conv, self.bn, W, b, gamma, beta)
        # VisualBackprop: the pooling node's input relu(bn1(conv1)) is never materialised -- its channel mean is taken from
        # the conv output and the BN coefficients (conv1's own tap needs the unpadded frames: SheepLocalizer records it)
        ops.vbp_tap(self.c, 3, 2, 0, st=self.st)
        if config.enable_backprop and config.train:     # the backward's sums read the conv values at the argmaxes (ops.POOL_ARGMAX_VALUES)
            y, self.idx, self.xsel = ops.bn_relu_maxpool(self.c, self.st, want_sel=True)
        else:
            (y, self.idx), self.xsel = ops.bn_relu_maxpool(self.c, self.st), None
        return y

    def backward(self, inputs, gys):
        _require_train()
        _, W, b, gamma, beta = self.inputs
        gc = ops.pool_bn_backward(gys[0].contiguous(), self.idx, self.c, self.st, gamma.data, gamma.grad_view,
                                  beta.grad_view, gbias=b.grad_view, xsel=self.xsel)
        ops.conv_wgrad(self.x, gc, W.grad_view, self.geo)
        return None, None, None, None, None

    def release(self):
        self.x = self.c = self.st = self.idx = self.xsel = None


class ResidualUnitFunction(Function):
    """relu( bn_n(conv_n( ... relu(bn_1(conv_1(x))) ... )) + shortcut(x) ), shortcut = bn_s(conv_s(x)) or x.

    One node for every residual unit on the path:
      BasicA / BasicB ............ sheep/resnet.py:136-141,156-160   (2 stages; 3x3-strided conv shortcut / identity)
      BottleNeckA / BottleNeckB .. sheep/resnet.py:163-216 = chainer ResNet50Layers' blocks (3 stages; 1x1 shortcut / identity)
      chainercv Bottleneck ....... Resnet50SheepLocalizer's res6 / res7 (sheep_localizer.py:131-132)
    ``stages`` = [(conv_link, bn_link), ...], ``shortcut`` = (conv_link, bn_link) or None.
    inputs: x, then (W, gamma, beta) per stage, then (W, gamma, beta) of the shortcut."""

    def __init__(self, stages, shortcut=None):
        self.stages, self.shortcut = list(stages), shortcut

    def forward(self, inputs):
        x = inputs[0]
        n = len(self.stages)
        self.x = x
        self.c, self.st, self.geo, self.h = [None] * n, [None] * n, [None] * n, [None] * n
        h = x
        paired = self._forward_pair(x, inputs) if self.shortcut is not None else False
        self.onload = [False] * n       # stage i reads relu(bn_{i-1}(c[i-1])) ON LOAD: h[i-1] is never materialised (ops.BN_ON_LOAD)
        for i, (conv, bn) in enumerate(self.stages):
            W, g, b = inputs[1 + 3 * i:4 + 3 * i]
            if h is not None:
                ops.vbp_tap(h, conv.ksize, conv.stride, conv.pad)       # main-branch convolution i and its input
            if self.onload[i]:
                geo = self.geo[i]
                stats = _zeros_stats(conv.out_channels, x.device)
                self.c[i] = ops.conv_fprop_affine(self.c[i - 1], self.st[i - 1], W, geo, stats=stats)
                self.st[i] = ops.bn_finalize(stats, geo.B * geo.Ho * geo.Wo, g, b, bn.avg_mean, bn.avg_var)
            elif not (paired and i == 0):
                self.c[i], self.st[i], self.geo[i] = _ConvBN.forward(h, conv, bn, W, None, g, b)
            if i < n - 1:
                # bn_i -> relu -> a 1 x 1 convolution the VGPR-fed kernels take (a bottleneck's bn2 -> conv3): the convolution and
                # its weight gradient apply the BN while they load its input, the backward takes the BN's sums from the data
                # gradient's epilogue and its mask from the recomputed sign -- nobody reads the activation
                nxt = self.stages[i + 1][0]
                if config.train and config.enable_backprop and ops.VBP_TAPS is None and not nxt.dense_rows:
                    gnext = nxt.geometry(self.c[i].shape[0], self.c[i].shape[1], self.c[i].shape[2])
                    if ops.affine_in_ok(gnext, self.c[i]) and ops.bn_sums_ok(gnext, self.c[i]):
                        self.geo[i + 1], self.onload[i + 1] = gnext, True
                        h = self.h[i] = None
                        continue
                h = self.h[i] = ops.bn_apply(self.c[i], self.st[i], relu=True)
        bits = config.train and config.enable_backprop     # the backward's ReLU mask as sign bits (ops.bn_apply)
        if self.shortcut is not None:
            W, g, b = inputs[1 + 3 * n:4 + 3 * n]
            if not paired:
                self.cs, self.sts, self.geos = _ConvBN.forward(x, self.shortcut[0], self.shortcut[1], W, None, g, b)
            self.out = ops.bn_apply(self.c[-1], self.st[-1], relu=True, x2=self.cs, st2=self.sts, want_bits=bits)
        else:
            self.out = ops.bn_apply(self.c[-1], self.st[-1], relu=True, residual=x, want_bits=bits)
        return self.out

    def _forward_pair(self, x, inputs):
        """the first stage's conv and the shortcut conv read the same x with the same kernel / stride / padding
        (BasicA: sheep/resnet.py:128-133; a bottleneck's conv1 / conv4): in training, fp32, they go out as ONE launch
        whose two tile sets share a grid and a tail (ops.conv_fprop_pair).  Sets c[0], st[0], geo[0], cs, sts, geos."""
        n = len(self.stages)
        (conv, bn), (convs, bns) = self.stages[0], self.shortcut
        B, H, Wd, _ = x.shape
        geo, geos = conv.geometry(B, H, Wd), convs.geometry(B, H, Wd)
        if not config.train or conv.dense_rows or not ops.fprop_pair_ok(x, geo, geos):
            return False
        W, g, b = inputs[1:4]
        Ws, gs, bs = inputs[1 + 3 * n:4 + 3 * n]
        sa, sb = _zeros_stats(conv.out_channels, x.device), _zeros_stats(convs.out_channels, x.device)
        self.c[0], self.cs = ops.conv_fprop_pair(x, W, Ws, geo, geos, sa, sb)
        self.st[0] = ops.bn_finalize(sa, B * geo.Ho * geo.Wo, g, b, bn.avg_mean, bn.avg_var)
        self.sts = ops.bn_finalize(sb, B * geos.Ho * geos.Wo, gs, bs, bns.avg_mean, bns.avg_var)
        self.geo[0], self.geos = geo, geos
        return True

    def backward(self, inputs, gys):
        _require_train()
        n = len(self.stages)
        P = self.inputs
        gout = gys[0].contiguous()
        Wl, gl, bl = P[1 + 3 * (n - 1):4 + 3 * (n - 1)]
        gcs = None
        if self.shortcut is not None:
            Ws, gs, bs = P[1 + 3 * n:4 + 3 * n]
            g, gcs = ops.bn_backward(gout, self.out, self.c[-1], self.st[-1], gl.data, gl.grad_view, bl.grad_view,
                                     x2=self.cs, st2=self.sts, gamma2=gs.data, ggamma2=gs.grad_view, gbeta2=bs.grad_view)
        else:
            g = ops.bn_backward(gout, self.out, self.c[-1], self.st[-1], gl.data, gl.grad_view, bl.grad_view)
        for i in range(n - 1, 0, -1):
            W, _, _ = P[1 + 3 * i:4 + 3 * i]
            # dgrad first: the weight gradient goes to the side stream and, enqueued after the dgrad, starts when
            # that finishes -- it then overlaps the HBM-bound BN passes that follow instead of fighting the dgrad
            # for the matrix pipes
            _, gp, bp = P[1 + 3 * (i - 1):4 + 3 * (i - 1)]
            if ops.bn_sums_ok(self.geo[i], self.c[i - 1]):
                # the two sums of BN i-1's backward ride in this dgrad's epilogue (its output tile meets the BN's input tile
                # there): the reduction pass over gh disappears, the BN backward is one pass
                gh, sums = ops.conv_dgrad(g, W.data, self.geo[i], bn_sums=(self.c[i - 1], self.st[i - 1]))
                if self.onload[i]:          # (the activation was never written: the weight gradient applies the BN on load too)
                    ops.conv_wgrad(self.c[i - 1], g, W.grad_view, self.geo[i], in_affine=self.st[i - 1])
                else:
                    ops.conv_wgrad(self.h[i - 1], g, W.grad_view, self.geo[i])
                g = ops.bn_backward_from_sums(gh, self.c[i - 1], self.st[i - 1], sums, gp.data, gp.grad_view, bp.grad_view)
                continue
            gh = ops.conv_dgrad(g, W.data, self.geo[i])
            ops.conv_wgrad(self.h[i - 1], g, W.grad_view, self.geo[i])
            g = ops.bn_backward(gh, self.h[i - 1], self.c[i - 1], self.st[i - 1], gp.data, gp.grad_view, bp.grad_view,
                                mask_is_own_relu=True)         # h[i-1] = relu(bn(c[i-1]))
        W0 = P[1]
        gx = None
        if P[0].requires_grad:
            if self.shortcut is not None:
                gx = ops.conv_dgrad(gcs, P[1 + 3 * n].data, self.geos)
                ops.conv_dgrad(g, W0.data, self.geo[0], out=gx, addend=gx)
            else:
                # gx = dgrad(g) + gout * (out > 0)     (identity shortcut through the final ReLU)
                gx = ops.conv_dgrad(g, W0.data, self.geo[0], addend=gout, addend_mask_ref=self.out)
        ops.conv_wgrad(self.x, g, W0.grad_view, self.geo[0])
        if self.shortcut is not None:
            ops.conv_wgrad(self.x, gcs, P[1 + 3 * n].grad_view, self.geos)
        return (gx,) + (None,) * (len(P) - 1)

    def release(self):
        self.x = self.c = self.h = self.st = self.out = self.cs = None


def residual_unit(x, stages, shortcut=None):
    args = [x]
    for conv, bn in list(stages) + ([shortcut] if shortcut is not None else []):
        args += [conv.W, bn.gamma, bn.beta]
    return ResidualUnitFunction(stages, shortcut)(*args)


# --------------------------------------------------------------------------- #
# assessor blocks (no BN, no bias; pre-activation)
# --------------------------------------------------------------------------- #
class DownResBlock1Function(Function):
    """c1(relu(c0(x))) + cs(x)   (common/net.py:19-25).  inputs: x, W0, W1, Ws."""

    def __init__(self, block):
        self.blk = block

    def forward(self, inputs):
        x, W0, W1, Ws = inputs
        k = self.blk
        B, H, Wd, _ = x.shape
        self.x = x
        self.g0 = k.c0.geometry(B, H, Wd)
        self.gs = k.cs.geometry(B, H, Wd)
        o16 = ops.STORAGE == 'bf16' and x.dtype == torch.float32      # the crops stay fp32; the bf16 region starts here
        self.h1 = ops.conv_fprop(x, W0, self.g0, out_bf16=o16)
        self.g1 = k.c1.geometry(B, self.g0.Ho, self.g0.Wo)
        out = ops.conv_fprop(x, Ws, self.gs, out_bf16=o16)
        ops.conv_fprop(self.h1, W1, self.g1, out=out, relu_in=True, addend=out)
        return out

    def backward(self, inputs, gys):
        xv, W0, W1, Ws = self.inputs
        g = gys[0].contiguous()
        wg = _wgrad_enabled(W0)
        need_gx = xv.requires_grad
        if not (wg or need_gx):
            return None, None, None, None
        gh1 = ops.conv_dgrad(g, W1.data, self.g1, mask_ref=self.h1)
        if wg:
            ops.conv_wgrad(self.h1, g, W1.grad_view, self.g1, relu_in=True)
            ops.conv_wgrad(self.x, g, Ws.grad_view, self.gs)
            ops.conv_wgrad(self.x, gh1, W0.grad_view, self.g0)
        gx = None
        if need_gx:
            if ops.crop_dgrad_ok(self.g0, self.gs, gh1, g):
                # the 4-channel crops: both data gradients in ONE launch that reads g and gh1 once (csrc/cropgrad.hip)
                gx = ops.crop_dgrad(gh1, W0.data, self.g0, g, Ws.data, self.gs)
            else:
                gx = ops.conv_dgrad(g, Ws.data, self.gs)
                ops.conv_dgrad(gh1, W0.data, self.g0, out=gx, addend=gx)
        return gx, None, None, None

    def release(self):
        self.x = self.h1 = None


class DownResBlock2Function(Function):
    """c1(relu(c0(relu(x)))) + cs(x)   (common/net.py:41-47).  inputs: x, W0, W1, Ws."""

    def __init__(self, block):
        self.blk = block

    def forward(self, inputs):
        x, W0, W1, Ws = inputs
        k = self.blk
        B, H, Wd, _ = x.shape
        self.x = x
        self.g0 = k.c0.geometry(B, H, Wd)
        self.gs = k.cs.geometry(B, H, Wd)
        self.h1 = ops.conv_fprop(x, W0, self.g0, relu_in=True)
        self.g1 = k.c1.geometry(B, self.g0.Ho, self.g0.Wo)
        out = ops.conv_fprop(x, Ws, self.gs)
        ops.conv_fprop(self.h1, W1, self.g1, out=out, relu_in=True, addend=out)
        return out

    def backward(self, inputs, gys):
        xv, W0, W1, Ws = self.inputs
        g = gys[0].contiguous()
        wg = _wgrad_enabled(W0)
        gh1 = ops.conv_dgrad(g, W1.data, self.g1, mask_ref=self.h1)
        if wg:
            ops.conv_wgrad(self.h1, g, W1.grad_view, self.g1, relu_in=True)
            ops.conv_wgrad(self.x, g, Ws.grad_view, self.gs)
            ops.conv_wgrad(self.x, gh1, W0.grad_view, self.g0, relu_in=True)
        gx = ops.conv_dgrad(g, Ws.data, self.gs)
        ops.conv_dgrad(gh1, W0.data, self.g0, out=gx, mask_ref=self.x, addend=gx)
        return gx, None, None, None

    def release(self):
        self.x = self.h1 = None


class DownResBlock3Function(Function):
    """c1(relu(c0(relu(x)))) + x   (common/net.py:62-67).  inputs: x, W0, W1."""

    def __init__(self, block):
        self.blk = block

    def forward(self, inputs):
        x, W0, W1 = inputs
        k = self.blk
        B, H, Wd, _ = x.shape
        self.x = x
        self.g0 = k.c0.geometry(B, H, Wd)
        self.g1 = k.c1.geometry(B, H, Wd)
        self.h1 = ops.conv_fprop(x, W0, self.g0, relu_in=True)
        return ops.conv_fprop(self.h1, W1, self.g1, relu_in=True, addend=x)

    def backward(self, inputs, gys):
        xv, W0, W1 = self.inputs
        g = gys[0].contiguous()
        wg = _wgrad_enabled(W0)
        gh1 = ops.conv_dgrad(g, W1.data, self.g1, mask_ref=self.h1)
        if wg:
            ops.conv_wgrad(self.h1, g, W1.grad_view, self.g1, relu_in=True)
        gx = ops.conv_dgrad(gh1, W0.data, self.g0, mask_ref=self.x, addend=g)
        if wg:
            ops.conv_wgrad(self.x, gh1, W0.grad_view, self.g0, relu_in=True)
        return gx, None, None

    def release(self):
        self.x = self.h1 = None
