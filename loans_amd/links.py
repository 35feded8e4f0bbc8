"""Parameter-holding links with the constructor signatures of the Chainer links
the reference instantiates: ``L.Convolution2D`` (sheep/resnet.py:43,128-133,
common/net.py:15-17), ``L.BatchNormalization`` (sheep/resnet.py:44),
``L.Linear`` (sheep_localizer.py:28, common/net.py:81).

They only own parameters (initialised on the host with NumPy's global RNG, like
Chainer's initialisers) and cached launch geometry; the arithmetic is in the
fused block functions (functions/blocks.py) so that BN statistics, ReLU and
residual sums ride in the conv kernels' epilogues.
"""
import numpy as np

from . import ops
from .runtime.core import Link, Parameter


# ---- initialisers (chainer.initializers.HeNormal / Normal / LeCunNormal / Constant) ----
class HeNormal:
    def __init__(self, scale=1.0, fan_option='fan_in'):
        self.scale, self.fan_option = scale, fan_option

    def __call__(self, shape):
        fan_in = int(np.prod(shape[1:]))
        fan_out = int(shape[0] * np.prod(shape[2:])) if len(shape) > 2 else shape[0]
        fan = fan_in if self.fan_option == 'fan_in' else fan_out
        return np.random.normal(0, self.scale * np.sqrt(2.0 / fan), shape).astype(np.float32)


class Normal:
    def __init__(self, scale=0.05):
        self.scale = scale

    def __call__(self, shape):
        return np.random.normal(0, self.scale, shape).astype(np.float32)


class LeCunNormal:
    def __call__(self, shape):
        fan_in = int(np.prod(shape[1:]))
        return np.random.normal(0, np.sqrt(1.0 / fan_in), shape).astype(np.float32)


def _pad4(c):
    return (c + 3) // 4 * 4


class Convolution2D(Link):
    """``L.Convolution2D(in_channels, out_channels, ksize, stride, pad, nobias, initialW)``.
    W is stored OHWI with the input channels padded to a multiple of 4."""

    def __init__(self, in_channels, out_channels, ksize=None, stride=1, pad=0, nobias=False, initialW=None,
                 dense_rows=False):
        super().__init__()
        if in_channels is None:
            raise ValueError('lazy in_channels is not supported: pass the channel count')
        self.in_channels, self.out_channels = in_channels, out_channels
        self.ksize, self.stride, self.pad = ksize, stride, pad
        self.cin_phys = _pad4(in_channels)
        # dense_rows (RGB stem): the input is the zero-padded packed-RGB frame buffer of ops.prep_images(..., geo)
        # and W is stored [Cout][kh][kwp][3] with zero weights on the kwp - kw window-padding pixels
        self.dense_rows = dense_rows
        init = initialW if initialW is not None else LeCunNormal()
        w = init((out_channels, in_channels, ksize, ksize))
        cin, cp = in_channels, self.cin_phys

        if dense_rows:
            assert in_channels == 3
            kwp = ops.dense_window(ksize)

            def to_logical(phys):
                return phys[:, :, :ksize, :].transpose(0, 3, 1, 2)

            def from_logical(a):
                out = np.zeros((a.shape[0], ksize, kwp, 3), np.float32)
                out[:, :, :ksize, :] = a.transpose(0, 2, 3, 1)
                return out
        else:
            def to_logical(phys):
                return phys.transpose(0, 3, 1, 2)[:, :cin]

            def from_logical(a):
                out = np.zeros((a.shape[0], a.shape[2], a.shape[3], cp), np.float32)
                out[..., :cin] = a.transpose(0, 2, 3, 1)
                return out

        with self.init_scope():
            self.W = Parameter(from_logical(w), w.shape, to_logical, from_logical)
            self.b = None if nobias else Parameter(np.zeros(out_channels, np.float32), (out_channels,))
        if nobias:
            self.__dict__['b'] = None
        self.__dict__['_geo'] = {}

    def geometry(self, B, H, W):
        key = (B, H, W)
        g = self._geo.get(key)
        if g is None:
            if self.dense_rows:
                g = ops.ConvGeometry(B, H, W, 3, self.out_channels, self.ksize, self.stride, self.pad, dense=True)
            else:
                g = ops.ConvGeometry(B, H, W, self.cin_phys, self.out_channels, self.ksize, self.stride, self.pad)
            self._geo[key] = g
        return g


class BatchNormalization(Link):
    """``L.BatchNormalization(size)``: gamma=1, beta=0, avg_mean=0, avg_var=1, decay .9, eps 2e-5."""

    def __init__(self, size, decay=0.9, eps=2e-5):
        super().__init__()
        assert decay == ops.BN_DECAY and eps == ops.BN_EPS, 'only the defaults the reference uses are built'
        self.size = size
        with self.init_scope():
            self.gamma = Parameter(np.ones(size, np.float32), (size,))
            self.beta = Parameter(np.zeros(size, np.float32), (size,))
        self.add_persistent('avg_mean', np.zeros(size, np.float32))
        self.add_persistent('avg_var', np.ones(size, np.float32))
        self.add_persistent('N', 0)


class Linear(Link):
    """``L.Linear(in_size, out_size, nobias, initialW)``.  ``in_size=None`` is resolved at the
    first call like Chainer.  ``nhwc_input=(H, W, C)`` declares that the kernels feed an
    NHWC-flattened activation, so W is stored in (h, w, c) order and exposed in Chainer's
    (c, h, w) order."""

    def __init__(self, in_size, out_size=None, nobias=False, initialW=None):
        super().__init__()
        if out_size is None:
            in_size, out_size = None, in_size
        self.out_size, self.nobias = out_size, nobias
        self._initialW = initialW if initialW is not None else LeCunNormal()
        self.__dict__['W'] = None
        self.__dict__['b'] = None
        if in_size is not None:
            self._initialize(in_size, None)

    def _initialize(self, in_size, nhwc_input):
        w = self._initialW((self.out_size, in_size))
        to_l = from_l = None
        phys = w
        if nhwc_input is not None:
            H, W, C = nhwc_input
            n = self.out_size

            def to_l(p):
                return p.reshape(n, H, W, C).transpose(0, 3, 1, 2).reshape(n, -1)

            def from_l(a):
                return a.reshape(n, C, H, W).transpose(0, 2, 3, 1).reshape(n, -1)

            phys = from_l(w)
        with self.init_scope():
            self.W = Parameter(phys, w.shape, to_l, from_l)
            if not self.nobias:
                self.b = Parameter(np.zeros(self.out_size, np.float32), (self.out_size,))
        self.in_size = in_size

    def ensure_initialized(self, in_size, nhwc_input=None):
        if self.W is None:
            if self._arena is not None:
                raise RuntimeError('Linear initialised after its model was finalised')
            self._initialize(in_size, nhwc_input)
