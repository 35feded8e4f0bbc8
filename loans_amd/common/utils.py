"""``Size`` and the two grid regularisers the updater uses (reference
common/utils.py:8, :137-178 DirectionLossCalculator, :301-316
OutOfImageLossCalculator).  Same classes, constructor (``xp``) and
``calc_loss(grids, image_size)`` signature; each loss is one fused HIP kernel
forward and one backward (the reference composes ~10 Chainer ops per loss).

The other calculators of the reference file (IOU / MinArea / MaxArea / ...,
:21-298) are not referenced by the training path."""
from collections import namedtuple

from ..functions.ops_small import GridLoss

Size = namedtuple('Size', ['height', 'width'])


class LossCalculator:

    def __init__(self, xp):
        self.xp = xp

    def calc_loss(self, grids, image_size):
        raise NotImplementedError


class DirectionLossCalculator(LossCalculator):
    """mean(max(TL_y - BL_y, 0)) + mean(max(TL_x - TR_x, 0)) on the image-scaled grid corners."""

    def calc_loss(self, grids, image_size):
        return GridLoss(0, img_h=image_size.height, img_w=image_size.width)(grids)


class OutOfImageLossCalculator(LossCalculator):
    """sum(|min(v + 1, 0)|) + sum(max(v - 1, 0)) over v = [TL_x, TL_y, TR_x, BL_y] of every sample.

    It is a SUM over the batch (common/utils.py:315).  Under data-parallel training the
    gradients of all ranks are averaged, so the term is scaled by ``batch_sum_scale``
    (= world size) to keep the global-batch gradient identical."""

    batch_sum_scale = 1.0

    def calc_loss(self, grids, image_size):
        return GridLoss(1, oob_scale=self.batch_sum_scale)(grids)
