"""``SheepLocalizer`` (reference sheep/sheep_localizer.py:18-117), MI355X-native.

Same constructor, call signature, attributes and ``predict`` contract:
``localizer(images (B,3,H,W) f32 RGB in [0,1]) -> (rois (B,3,th,tw), points (B,2,th,tw))``.

What changed underneath: ``prepare_images`` is one HIP kernel instead of a
device->host->PIL->device round trip per image (:72-82); the backbone runs NHWC
on the fp32 MFMA implicit-GEMM kernels; ``rois`` is an NCHW view of the NHWC4
buffer the assessor consumes directly.
"""
import numpy as np
import torch

from .. import links as L
from .. import ops
from ..common.utils import Size
from ..functions import (global_average_pooling_2d, linear, reshape, rotation_dropout,
                         spatial_transformer_grid, spatial_transformer_sampler)
from ..functions.ops_small import ExposeNCHW, rois_to_grayscale
from ..runtime.core import Chain, Variable, as_variable, config, using_config
from ..chainercv_resnet import ResBlock
from ..iou.iou_regressor import MyResNet50Layers
from .resnet import BasicBlock, ResNet


def _as_device_batch(images, device):
    if isinstance(images, Variable):
        images = images.data
    if isinstance(images, np.ndarray):
        images = torch.from_numpy(np.ascontiguousarray(images, dtype=np.float32))
    return images.to(device=device, dtype=torch.float32).contiguous()


class SheepLocalizer(Chain):

    def __init__(self, out_size, transform_rois_to_grayscale=False, train_imagenet=False):
        super().__init__()
        if train_imagenet:
            raise NotImplementedError("ImageNet pre-training head is outside the LoANs training path")
        with self.init_scope():
            self.feature_extractor = ResNet(18, class_labels=None)
            self.res6 = BasicBlock(2, 512, in_ch=512)
            self.res7 = BasicBlock(2, 512, in_ch=512)
            self.param_predictor = L.Linear(512, 6)

            transform_bias = self.param_predictor.b.host
            transform_bias[[0, 4]] = 0.8
            transform_bias[[2, 5]] = 0
            self.param_predictor.W.host[...] = 0

        self.cold_links = ('res6', 'res7')          # arena tail: only used on frames taller than 224 / 300 px
        self.visual_backprop_anchors = []
        self.out_size = tuple(out_size)
        self.transform_rois_to_grayscale = transform_rois_to_grayscale
        self.train_imagenet = train_imagenet

    def __call__(self, images):
        self.visual_backprop_anchors.clear()
        device = images.data.device if isinstance(images, Variable) else (
            images.device if torch.is_tensor(images) and images.is_cuda else torch.device('cuda', torch.cuda.current_device()))
        self.finalize(device)
        images = _as_device_batch(images, device)
        height = images.shape[-2]
        self.arena.set_active('res6' if height <= 224 else ('res7' if height <= 300 else None))

        input_images = self.prepare_images(images)
        self._record_stem_tap(images)
        h = self.feature_extractor(input_images)

        if images.shape[-2] > 224:
            h = self.res6(h)

            if images.shape[-2] > 300:
                h = self.res7(h)

        self._anchor(h)
        h = global_average_pooling_2d(h)

        transform_params = linear(h, self.param_predictor.W, self.param_predictor.b)
        transform_params = rotation_dropout(reshape(transform_params, (-1, 2, 3)), ratio=0.0)
        points = spatial_transformer_grid(transform_params, self.out_size)
        rois = ExposeNCHW()(spatial_transformer_sampler(as_variable(images), points))
        self.last_transform_params = transform_params

        if self.transform_rois_to_grayscale:
            rois = rois_to_grayscale(rois)

        return rois, points

    def _record_stem_tap(self, images):
        """VisualBackprop: conv1's node -- the channel mean of the PREPROCESSED frames (the array conv1 reads in the reference,
        sheep_localizer.py:45-46) at the frame's own size; the stem's fused function only sees the zero-padded buffer"""
        if ops.VBP_TAPS is not None:
            conv1 = self.feature_extractor.conv1
            ops.vbp_tap(ops.prep_images(images), conv1.ksize, conv1.stride, conv1.pad, cdiv=3)

    def _anchor(self, h):
        if ops.VBP_TAPS is not None:
            h.vbp_taps = list(ops.VBP_TAPS)
        self.visual_backprop_anchors.append(h)

    def prepare_images(self, images):
        """``images * 255`` -> uint8 truncation -> BGR -> minus mean, cutting the graph
        (sheep_localizer.py:45,72-82); returns the zero-padded packed-RGB buffer conv1's dense K rows read."""
        B, _, H, W = images.shape
        return Variable(ops.prep_images(images, self.feature_extractor.conv1.geometry(B, H, W)), requires_grad=False)

    def extract_corners(self, bboxes):
        data = bboxes.data if isinstance(bboxes, Variable) else bboxes
        top = data[:, 1, 0, 0]
        left = data[:, 0, 0, 0]
        bottom = data[:, 1, -1, -1]
        right = data[:, 0, -1, -1]
        return torch.stack([top, left, bottom, right], dim=1)

    def scale_bboxes(self, bboxes, image_size):
        bboxes = (bboxes + 1) / 2
        bboxes[:, ::2] *= image_size.height
        bboxes[:, 1::2] *= image_size.width
        return bboxes

    def predict(self, images, return_visual_backprop=False):
        images = np.stack([np.asarray(image, dtype=np.float32) for image in images], axis=0)
        with using_config('train', False), using_config('enable_backprop', False):
            old_taps, ops.VBP_TAPS = ops.VBP_TAPS, ([] if return_visual_backprop else None)
            try:
                rois, bboxes = self(images)
            finally:
                ops.VBP_TAPS = old_taps
            if return_visual_backprop:
                if not hasattr(self, 'visual_backprop'):
                    from ..insights.visual_backprop import VisualBackprop
                    self.visual_backprop = VisualBackprop()
                visual_backprop = self.visual_backprop.perform_visual_backprop(self.visual_backprop_anchors[0]).cpu().numpy()
            else:
                visual_backprop = None
            bboxes = self.extract_corners(bboxes)
            bboxes = self.scale_bboxes(bboxes, Size._make(images.shape[-2:]))

        bboxes = [bbox.cpu().numpy().reshape(1, -1) for bbox in bboxes]

        return bboxes, rois, np.ones((len(bboxes), 1)), visual_backprop


class Resnet50SheepLocalizer(SheepLocalizer):
    """ResNet-50 localizer (reference sheep/sheep_localizer.py:120-178): Chainer ``ResNet50Layers`` backbone
    (taps ``res5`` / ``pool5``), chainercv ``ResBlock`` stages res6 / res7 on frames taller than 224 / 300 px,
    ``Linear(2048, 6)``, the same STN tail."""

    def __init__(self, out_size, transform_rois_to_grayscale=False, train_imagenet=False):
        super(SheepLocalizer, self).__init__()
        if train_imagenet:
            raise NotImplementedError("ImageNet pre-training head is outside the LoANs training path")
        initialW = L.HeNormal(scale=1., fan_option='fan_out')
        keys_to_remove = ['fc6', 'prob']
        with self.init_scope():
            self.feature_extractor = MyResNet50Layers(keys_to_remove=keys_to_remove, pretrained_model='auto')
            self.param_predictor = L.Linear(2048, 6)

            self.res6 = ResBlock(2, 2048, 1024, 2048, 2, initialW=initialW)
            self.res7 = ResBlock(2, 2048, 1024, 2048, 2, initialW=initialW)

            transform_bias = self.param_predictor.b.host
            transform_bias[[0, 4]] = 0.8
            transform_bias[[2, 5]] = 0
            self.param_predictor.W.host[...] = 0

        self.cold_links = ('res6', 'res7', 'feature_extractor/fc6')
        self.visual_backprop_anchors = []
        self.out_size = tuple(out_size)
        self.transform_rois_to_grayscale = transform_rois_to_grayscale
        self.train_imagenet = train_imagenet

    def __call__(self, images):
        self.visual_backprop_anchors.clear()
        device = images.data.device if isinstance(images, Variable) else (
            images.device if torch.is_tensor(images) and images.is_cuda else torch.device('cuda', torch.cuda.current_device()))
        self.finalize(device)
        images = _as_device_batch(images, device)
        height = images.shape[-2]
        self.arena.set_active('res6' if height <= 224 else ('res7' if height <= 300 else 'feature_extractor/fc6'))

        input_images = self.prepare_images(images)
        self._record_stem_tap(images)
        h = self.feature_extractor(input_images, layers=['res5', 'pool5'])

        self._anchor(h['res5'])
        if images.shape[-2] > 224:
            h = h['res5']
            h = self.res6(h)

            if images.shape[-2] > 300:
                h = self.res7(h)

            h = global_average_pooling_2d(h)
        else:
            h = h['pool5']

        transform_params = linear(h, self.param_predictor.W, self.param_predictor.b)
        transform_params = rotation_dropout(reshape(transform_params, (-1, 2, 3)), ratio=0.0)
        points = spatial_transformer_grid(transform_params, self.out_size)
        rois = ExposeNCHW()(spatial_transformer_sampler(as_variable(images), points))
        self.last_transform_params = transform_params

        if self.transform_rois_to_grayscale:
            rois = rois_to_grayscale(rois)

        return rois, points
