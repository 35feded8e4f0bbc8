"""Rotation dropout — the reference's only user-defined operator
(functions/rotation_droput.py:9-52), same class name, constructor and
forward/backward contract; the multiply runs in the HIP ``loans_mul_f32`` kernel.

With ``ratio=0.0`` (how sheep_localizer.py:61 calls it) theta[:,0,1] and
theta[:,1,0] are zeroed in both train and test mode; train mode consumes one
``numpy.random.rand(1)`` draw exactly like the reference."""
import numpy
import torch

from .. import ops
from ..runtime.core import Function, config


class RotationDropout(Function):

    def __init__(self, dropout_ratio):
        self.dropout_ratio = dropout_ratio

    def check_type_forward(self, x):
        if not (x.dtype.is_floating_point and x.dim() == 3 and x.shape[1] == 2 and x.shape[2] == 3):
            raise TypeError('rotation_dropout expects a float array of shape (B, 2, 3), got %s' % (tuple(x.shape),))

    def _mask(self, x, flag):
        mask = torch.ones_like(x)
        mask[:, 0, 1] = flag
        mask[:, 1, 0] = flag
        return mask

    def forward(self, x):
        self.check_type_forward(x[0])
        if not config.train:
            # scale affected weights if we are testing (rotation_droput.py:30-36)
            self.mask = self._mask(x[0], float(self.dropout_ratio))
            return ops.mul(x[0].contiguous(), self.mask, keep=True)
        if not hasattr(self, 'mask'):
            flag_data = bool(numpy.random.rand(1)[0] < self.dropout_ratio)      # :41
            self.mask = self._mask(x[0], float(flag_data))
        # keep: theta is what SheepLocalizer.last_transform_params holds across steps (ops._StepArena's contract)
        return ops.mul(x[0].contiguous(), self.mask, keep=True)

    def backward(self, x, gy):
        return ops.mul(gy[0].contiguous(), self.mask)


def rotation_dropout(x, ratio=.5, **kwargs):
    return RotationDropout(ratio)(x)
