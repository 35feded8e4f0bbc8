"""Differentiable wrappers of the small kernels, named after the chainer.functions
the reference calls: ``_global_average_pooling_2d`` (sheep_localizer.py:58),
``L.Linear`` (:60), ``F.spatial_transformer_grid`` / ``_sampler`` (:62-63),
``F.mean_squared_error`` (sheep_updater.py:43,60), ``sigmoid(l4(relu(h)))``
(common/net.py:89-90)."""
import torch

from .. import ops
from ..runtime.core import Function, Parameter, Variable


class GlobalAveragePooling2D(Function):
    """x: NHWC (B,H,W,C) -> (B,C)."""

    def forward(self, inputs):
        self.shape, self.dtype = tuple(inputs[0].shape), inputs[0].dtype
        return ops.gap_fwd(inputs[0])

    def backward(self, inputs, gys):
        return ops.gap_bwd(gys[0].contiguous(), self.shape, self.dtype)


def global_average_pooling_2d(x):
    return GlobalAveragePooling2D()(x)


class LinearFunction(Function):
    """y = act_out(act_in(x) W^T + b); accumulates gW / gb into the arena itself."""

    def __init__(self, act_in=False, act_out=False):
        self.act_in, self.act_out = act_in, act_out

    def forward(self, inputs):
        x, W = inputs[0], inputs[1]
        b = inputs[2] if len(inputs) > 2 else None
        self.y = ops.linear_fwd(x, W, b, self.act_in, self.act_out)
        return self.y

    def backward(self, inputs, gys):
        x, W = inputs[0], inputs[1]
        Wp = self.inputs[1]
        bp = self.inputs[2] if len(inputs) > 2 else None
        wgrad = Wp.update_rule.enabled or not getattr(Wp, 'skip_grad_when_disabled', False)
        gx = ops.linear_bwd(x, W, self.y, gys[0].contiguous(),
                            gW=Wp.grad_view if wgrad else None,
                            gb=bp.grad_view if (bp is not None and wgrad) else None,
                            need_gx=self.inputs[0].requires_grad, act_in=self.act_in, act_out=self.act_out)
        return (gx,) + (None,) * (len(inputs) - 1)

    def release(self):
        self.y = None


def linear(x, W, b=None):
    args = (x, W) if b is None else (x, W, b)
    return LinearFunction()(*args)


def sigmoid_linear_head(x, W):
    """sigmoid(Linear(relu(x))) fused (common/net.py:89-90)."""
    return LinearFunction(act_in=True, act_out=True)(x, W)


class SpatialTransformerGrid(Function):
    def __init__(self, output_shape):
        self.output_shape = tuple(output_shape)

    def forward(self, inputs):
        return ops.st_grid_fwd(inputs[0].contiguous(), self.output_shape)

    def backward(self, inputs, gys):
        return ops.st_grid_bwd(gys[0].contiguous())


def spatial_transformer_grid(theta, output_shape):
    return SpatialTransformerGrid(output_shape)(theta)


class SpatialTransformerSampler(Function):
    """images: NCHW leaf (B,3,H,W); grid (B,2,th,tw) -> rois as an NHWC4 buffer exposed
    through a (B,3,th,tw) view.  Only the grid gradient exists on this path (the frames
    are data, sheep_localizer.py:63)."""

    def forward(self, inputs):
        self.images, self.grid = inputs[0], inputs[1].contiguous()
        self.rois_nhwc4 = ops.st_sampler_fwd(self.images, self.grid)
        return self.rois_nhwc4

    def backward(self, inputs, gys):
        return None, ops.st_sampler_bwd_grid(self.images, self.grid, gys[0].contiguous())

    def release(self):
        self.images = self.grid = self.rois_nhwc4 = None


def spatial_transformer_sampler(images, grid):
    return SpatialTransformerSampler()(images, grid)


class MeanSquaredError(Function):
    def __init__(self, tconst=None):
        self.tconst = tconst

    def forward(self, inputs):
        y = inputs[0].contiguous()
        t = inputs[1].contiguous() if len(inputs) > 1 else None
        return ops.mse_fwd(y, t, 0.0 if self.tconst is None else self.tconst)

    def backward(self, inputs, gys):
        y = inputs[0].contiguous()
        t = inputs[1].contiguous() if len(inputs) > 1 else None
        g = ops.mse_bwd(y, gys[0], t, 0.0 if self.tconst is None else self.tconst)
        return (g,) if t is None else (g, None)


def mean_squared_error(x0, x1):
    """``F.mean_squared_error(y, t)``.  ``t`` may be a constant-filled array created with
    ``xp.full`` (sheep_updater.py:42); any array works."""
    return MeanSquaredError()(x0, x1)


class GridLoss(Function):
    def __init__(self, kind, img_h=0.0, img_w=0.0, oob_scale=1.0):
        self.kind, self.img_h, self.img_w, self.oob_scale = kind, float(img_h), float(img_w), float(oob_scale)

    def forward(self, inputs):
        self.grid = inputs[0].contiguous()
        return ops.grid_loss_fwd(self.grid, self.kind, self.img_h, self.img_w, self.oob_scale)

    def backward(self, inputs, gys):
        return ops.grid_loss_bwd(self.grid, gys[0], self.kind, self.img_h, self.img_w, self.oob_scale)

    def release(self):
        self.grid = None


class ViewAsNHWC4(Function):
    """Re-interprets the NCHW *view* of an NHWC4 buffer (the localizer's ``rois``) as that
    buffer, and its gradient back; no data movement."""

    def forward(self, inputs):
        t = inputs[0]
        B, _, h, w = t.shape
        return torch.as_strided(t, (B, h, w, 4), (h * w * 4, w * 4, 4, 1))

    def backward(self, inputs, gys):
        return nchw_view(gys[0])


def nchw_view(nhwc4):
    """(B,h,w,4) buffer -> logical (B,3,h,w) view, the shape the reference exposes."""
    return nhwc4[..., :3].permute(0, 3, 1, 2)


class ExposeNCHW(Function):
    """NHWC4 -> logical (B,3,h,w) view (forward) and the inverse for the gradient."""

    def forward(self, inputs):
        return nchw_view(inputs[0])

    def backward(self, inputs, gys):
        g = gys[0]
        B, _, h, w = g.shape
        if g.stride() == (h * w * 4, 1, w * 4, 4):
            return torch.as_strided(g, (B, h, w, 4), (h * w * 4, w * 4, 4, 1))
        out = torch.zeros((B, h, w, 4), device=g.device, dtype=g.dtype)
        out[..., :3] = g.permute(0, 2, 3, 1)
        return out


class RoisToGrayscale(Function):
    """``0.299 * r + 0.587 * g + 0.114 * b`` with ``b, g, r = split_axis(rois, 3, axis=1)`` (reference
    sheep/sheep_localizer.py:65-68, ``transform_rois_to_grayscale=True``): (B,3,h,w) -> (B,1,h,w).  The rois are the NCHW
    view of the sampler's NHWC4 buffer; forward and backward are one streaming kernel each."""

    def forward(self, inputs):
        x = inputs[0]
        B, c, h, w = x.shape
        assert c == 3, "rois are not in RGB, can not convert them to grayscale"
        if x.stride() == (h * w * 4, 1, w * 4, 4):
            nhwc4 = torch.as_strided(x, (B, h, w, 4), (h * w * 4, w * 4, 4, 1))
        else:
            nhwc4 = ops.nchw3_to_nhwc4(x.contiguous())
        return ops.gray_fwd(nhwc4).view(B, 1, h, w)

    def backward(self, inputs, gys):
        g = gys[0].contiguous()
        B, _, h, w = g.shape
        return nchw_view(ops.gray_bwd(g.view(B, h, w)))


def rois_to_grayscale(rois):
    return RoisToGrayscale()(rois)
