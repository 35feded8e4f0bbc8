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
        B, H, Wd, _ = x.shape
        geo = conv.geometry(B, H, Wd)
        if config.train:
            stats = _zeros_stats(conv.out_channels, x.device)
            c = ops.conv_fprop(x, W, geo, bias=b, stats=stats)
            st = ops.bn_finalize(stats, B * geo.Ho * geo.Wo, gamma, beta, bn.avg_mean, bn.avg_var)
        else:
            c = ops.conv_fprop(x, W, geo, bias=b)
            st = ops.bn_eval_coeffs(gamma, beta, bn.avg_mean, bn.avg_var)
        return c, st, geo


def _require_train():
    if not config.train:
        raise RuntimeError('backward through test-mode BatchNormalization is not on the training path')


class StemFunction(Function):
    """conv1(7x7/2, bias) -> bn1 -> relu -> max_pool(3, 2, cover_all)   (sheep/resnet.py:72-73).
    inputs: x (NHWC4 preprocessed frames, no gradient), W, b, gamma, beta."""

    def __init__(self, conv, bn):
        self.conv, self.bn = conv, bn

    def forward(self, inputs):
        x, W, b, gamma, beta = inputs
        self.x = x
        self.c, self.st, self.geo = _ConvBN.forward(x, self.conv, self.bn, W, b, gamma, beta)
        y, self.idx = ops.bn_relu_maxpool(self.c, self.st)
        return y

    def backward(self, inputs, gys):
        _require_train()
        _, W, b, gamma, beta = self.inputs
        g = ops.maxpool_relu_bwd(gys[0].contiguous(), self.idx, self.c, self.st)
        gc = ops.bn_backward(g, None, self.c, self.st, gamma.data, gamma.grad_view, beta.grad_view)
        ops.conv_wgrad(self.x, gc, W.grad_view, self.geo)
        ops.colsum_acc(gc, b.grad_view)
        return None, None, None, None, None

    def release(self):
        self.x = self.c = self.st = self.idx = None


class BasicAFunction(Function):
    """relu(bn2(conv2(relu(bn1(conv1(x))))) + bn3(conv3(x)))   (sheep/resnet.py:136-141).
    inputs: x, W1, g1, b1, W2, g2, b2, W3, g3, b3."""

    def __init__(self, block):
        self.blk = block

    def forward(self, inputs):
        x, W1, g1, b1, W2, g2, b2, W3, g3, b3 = inputs
        k = self.blk
        self.x = x
        self.c1, self.st1, self.geo1 = _ConvBN.forward(x, k.conv1, k.bn1, W1, None, g1, b1)
        self.h1 = ops.bn_apply(self.c1, self.st1, relu=True)
        self.c2, self.st2, self.geo2 = _ConvBN.forward(self.h1, k.conv2, k.bn2, W2, None, g2, b2)
        self.c3, self.st3, self.geo3 = _ConvBN.forward(x, k.conv3, k.bn3, W3, None, g3, b3)
        self.out = ops.bn_apply(self.c2, self.st2, relu=True, x2=self.c3, st2=self.st3)
        return self.out

    def backward(self, inputs, gys):
        _require_train()
        xv, W1, g1, b1, W2, g2, b2, W3, g3, b3 = self.inputs
        gout = gys[0].contiguous()
        gc2, gc3 = ops.bn_backward(gout, self.out, self.c2, self.st2, g2.data, g2.grad_view, b2.grad_view,
                                   x2=self.c3, st2=self.st3, gamma2=g3.data, ggamma2=g3.grad_view, gbeta2=b3.grad_view)
        ops.conv_wgrad(self.h1, gc2, W2.grad_view, self.geo2)
        ops.conv_wgrad(self.x, gc3, W3.grad_view, self.geo3)
        gh1 = ops.conv_dgrad(gc2, W2.data, self.geo2)
        gc1 = ops.bn_backward(gh1, self.h1, self.c1, self.st1, g1.data, g1.grad_view, b1.grad_view)
        ops.conv_wgrad(self.x, gc1, W1.grad_view, self.geo1)
        gx = None
        if xv.requires_grad:
            gx = ops.conv_dgrad(gc3, W3.data, self.geo3)
            ops.conv_dgrad(gc1, W1.data, self.geo1, out=gx, addend=gx)
        return (gx,) + (None,) * 9

    def release(self):
        self.x = self.c1 = self.c2 = self.c3 = self.h1 = self.out = None


class BasicBFunction(Function):
    """relu(bn2(conv2(relu(bn1(conv1(x))))) + x)   (sheep/resnet.py:156-160).
    inputs: x, W1, g1, b1, W2, g2, b2."""

    def __init__(self, block):
        self.blk = block

    def forward(self, inputs):
        x, W1, g1, b1, W2, g2, b2 = inputs
        k = self.blk
        self.x = x
        self.c1, self.st1, self.geo1 = _ConvBN.forward(x, k.conv1, k.bn1, W1, None, g1, b1)
        self.h1 = ops.bn_apply(self.c1, self.st1, relu=True)
        self.c2, self.st2, self.geo2 = _ConvBN.forward(self.h1, k.conv2, k.bn2, W2, None, g2, b2)
        self.out = ops.bn_apply(self.c2, self.st2, relu=True, residual=x)
        return self.out

    def backward(self, inputs, gys):
        _require_train()
        xv, W1, g1, b1, W2, g2, b2 = self.inputs
        gout = gys[0].contiguous()
        gc2 = ops.bn_backward(gout, self.out, self.c2, self.st2, g2.data, g2.grad_view, b2.grad_view)
        ops.conv_wgrad(self.h1, gc2, W2.grad_view, self.geo2)
        gh1 = ops.conv_dgrad(gc2, W2.data, self.geo2)
        gc1 = ops.bn_backward(gh1, self.h1, self.c1, self.st1, g1.data, g1.grad_view, b1.grad_view)
        ops.conv_wgrad(self.x, gc1, W1.grad_view, self.geo1)
        # gx = dgrad(gc1) + gout * (out > 0)     (identity shortcut through the final ReLU)
        gx = ops.conv_dgrad(gc1, W1.data, self.geo1, addend=gout, addend_mask_ref=self.out)
        return (gx,) + (None,) * 6

    def release(self):
        self.x = self.c1 = self.c2 = self.h1 = self.out = None


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
        self.h1 = ops.conv_fprop(x, W0, self.g0)
        self.g1 = k.c1.geometry(B, self.g0.Ho, self.g0.Wo)
        out = ops.conv_fprop(x, Ws, self.gs)
        ops.conv_fprop(self.h1, W1, self.g1, out=out, relu_in=True, addend=out)
        return out

    def backward(self, inputs, gys):
        xv, W0, W1, Ws = self.inputs
        g = gys[0].contiguous()
        wg = _wgrad_enabled(W0)
        if wg:
            ops.conv_wgrad(self.h1, g, W1.grad_view, self.g1, relu_in=True)
            ops.conv_wgrad(self.x, g, Ws.grad_view, self.gs)
        need_gx = xv.requires_grad
        if not (wg or need_gx):
            return None, None, None, None
        gh1 = ops.conv_dgrad(g, W1.data, self.g1, mask_ref=self.h1)
        if wg:
            ops.conv_wgrad(self.x, gh1, W0.grad_view, self.g0)
        gx = None
        if need_gx:
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
        if wg:
            ops.conv_wgrad(self.h1, g, W1.grad_view, self.g1, relu_in=True)
            ops.conv_wgrad(self.x, g, Ws.grad_view, self.gs)
        gh1 = ops.conv_dgrad(g, W1.data, self.g1, mask_ref=self.h1)
        if wg:
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
        if wg:
            ops.conv_wgrad(self.h1, g, W1.grad_view, self.g1, relu_in=True)
        gh1 = ops.conv_dgrad(g, W1.data, self.g1, mask_ref=self.h1)
        if wg:
            ops.conv_wgrad(self.x, gh1, W0.grad_view, self.g0, relu_in=True)
        gx = ops.conv_dgrad(gh1, W0.data, self.g0, mask_ref=self.x, addend=g)
        return gx, None, None

    def release(self):
        self.x = self.h1 = None
