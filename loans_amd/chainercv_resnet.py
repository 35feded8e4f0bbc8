"""``chainercv.links.model.resnet.ResBlock`` (chainercv 0.9.0), which the reference uses for the extra
stages ``res6`` / ``res7`` of ``Resnet50SheepLocalizer`` (sheep/sheep_localizer.py:9,131-132): bottleneck
``a`` with a 1x1 ``residual_conv`` + (n_layer - 1) identity bottlenecks ``b1`` ..., stride on the 3x3 conv
(``stride_first=False``).  Parameter paths match chainercv's (``a/conv1/conv/W``, ``a/conv1/bn/gamma`` ...)."""
from . import links as L
from .functions import blocks
from .runtime.core import Chain


class Conv2DBNActiv(Chain):
    def __init__(self, in_channels, out_channels, ksize, stride=1, pad=0, nobias=True, initialW=None):
        super().__init__()
        with self.init_scope():
            self.conv = L.Convolution2D(in_channels, out_channels, ksize, stride, pad, nobias=nobias, initialW=initialW)
            self.bn = L.BatchNormalization(out_channels)


class Bottleneck(Chain):
    def __init__(self, in_channels, mid_channels, out_channels, stride=1, initialW=None, residual_conv=False,
                 stride_first=False):
        super().__init__()
        s1, s3 = (stride, 1) if stride_first else (1, stride)
        with self.init_scope():
            self.conv1 = Conv2DBNActiv(in_channels, mid_channels, 1, s1, 0, initialW=initialW)
            self.conv2 = Conv2DBNActiv(mid_channels, mid_channels, 3, s3, 1, initialW=initialW)
            self.conv3 = Conv2DBNActiv(mid_channels, out_channels, 1, 1, 0, initialW=initialW)
            if residual_conv:
                self.residual_conv = Conv2DBNActiv(in_channels, out_channels, 1, stride, 0, initialW=initialW)
        self.has_residual_conv = residual_conv

    def __call__(self, x):
        stages = [(c.conv, c.bn) for c in (self.conv1, self.conv2, self.conv3)]
        sc = (self.residual_conv.conv, self.residual_conv.bn) if self.has_residual_conv else None
        return blocks.residual_unit(x, stages, sc)


class ResBlock(Chain):
    def __init__(self, n_layer, in_channels, mid_channels, out_channels, stride, initialW=None, stride_first=False):
        super().__init__()
        if in_channels is None:
            raise ValueError('pass in_channels explicitly (the reference relies on lazy initialisation)')
        with self.init_scope():
            self.a = Bottleneck(in_channels, mid_channels, out_channels, stride, initialW, residual_conv=True,
                                stride_first=stride_first)
            self._forward = ['a']
            for i in range(n_layer - 1):
                name = 'b{}'.format(i + 1)
                setattr(self, name, Bottleneck(out_channels, mid_channels, out_channels, 1, initialW))
                self._forward.append(name)

    def __call__(self, x):
        for name in self._forward:
            x = getattr(self, name)(x)
        return x
