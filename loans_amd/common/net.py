"""The assessor / IoU regressor (reference common/net.py:6-90): BN-free
pre-activation residual CNN, ``Linear(None, 1, nobias)`` -> sigmoid.  Same class
names, constructor arguments and parameter paths (``r0/c0/W`` ... ``l4/W``)."""
import torch

from .. import links as L
from .. import ops
from ..functions import blocks
from ..functions.ops_small import sigmoid_linear_head
from ..runtime.core import Chain, Variable, as_variable


class DownResBlock1(Chain):
    """pre activation residual block"""

    def __init__(self, ch, in_ch=3):
        w = L.Normal(0.02)
        super(DownResBlock1, self).__init__()
        with self.init_scope():
            self.c0 = L.Convolution2D(in_ch, ch, 3, 1, 1, initialW=w, nobias=True)
            self.c1 = L.Convolution2D(ch, ch, 4, 2, 1, initialW=w, nobias=True)
            self.cs = L.Convolution2D(in_ch, ch, 4, 2, 1, initialW=w, nobias=True)

    def __call__(self, x):
        return blocks.DownResBlock1Function(self)(x, self.c0.W, self.c1.W, self.cs.W)


class DownResBlock2(Chain):
    """pre activation residual block"""

    def __init__(self, ch):
        w = L.Normal(0.02)
        super(DownResBlock2, self).__init__()
        with self.init_scope():
            self.c0 = L.Convolution2D(ch, ch, 3, 1, 1, initialW=w, nobias=True)
            self.c1 = L.Convolution2D(ch, ch, 4, 2, 1, initialW=w, nobias=True)
            self.cs = L.Convolution2D(ch, ch, 4, 2, 1, initialW=w, nobias=True)

    def __call__(self, x):
        return blocks.DownResBlock2Function(self)(x, self.c0.W, self.c1.W, self.cs.W)


class DownResBlock3(Chain):
    """pre activation residual block"""

    def __init__(self, ch):
        w = L.Normal(0.02)
        super(DownResBlock3, self).__init__()
        with self.init_scope():
            self.c0 = L.Convolution2D(ch, ch, 3, 1, 1, initialW=w, nobias=True)
            self.c1 = L.Convolution2D(ch, ch, 3, 1, 1, initialW=w, nobias=True)

    def __call__(self, x):
        return blocks.DownResBlock3Function(self)(x, self.c0.W, self.c1.W)


def nhwc4_of(x):
    """Accepts what the reference's callers hand the assessor -- an NCHW (B,3,h,w) batch --
    and returns the NHWC4 variable the kernels consume: zero-copy when ``x`` is the
    localizer's ``rois`` (an NCHW *view* of an NHWC4 buffer), one layout kernel otherwise."""
    x = as_variable(x)
    t = x.data
    if t.dim() != 4 or t.shape[1] != 3:
        raise ValueError('assessor input must be (B, 3, h, w), got %s' % (tuple(t.shape),))
    B, _, h, w = t.shape
    if t.stride() == (h * w * 4, 1, w * 4, 4):
        from ..functions.ops_small import ViewAsNHWC4
        return ViewAsNHWC4()(x)
    return Variable(ops.nchw3_to_nhwc4(t.contiguous()), requires_grad=False)


class ResnetAssessor(Chain):
    def __init__(self, bottom_width=8, ch=128, wscale=0.02, output_dim=1):
        w = L.Normal(wscale)
        super(ResnetAssessor, self).__init__()
        self.bottom_width = bottom_width
        self.ch = ch
        with self.init_scope():
            self.r0 = DownResBlock1(128)
            self.r1 = DownResBlock2(128)
            self.r2 = DownResBlock3(128)
            self.r3 = DownResBlock3(128)
            self.l4 = L.Linear(None, output_dim, initialW=w, nobias=True)

    def __call__(self, x):
        x = nhwc4_of(x)
        if self.l4.W is None:
            _, h, w, _ = x.shape
            for _ in range(2):
                h, w = ops.conv_outsize(h, 4, 2, 1), ops.conv_outsize(w, 4, 2, 1)
            self.l4.ensure_initialized(128 * h * w, nhwc_input=(h, w, 128))
        self.finalize(x.data.device)
        self.x = x
        self.h1 = self.r0(self.x)
        self.h2 = self.r1(self.h1)
        self.h3 = self.r2(self.h2)
        self.h4 = self.r3(self.h3)
        return sigmoid_linear_head(self.h4, self.l4.W)
