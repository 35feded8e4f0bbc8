"""``MyResNet50Layers`` (reference iou/iou_regressor.py:4-15) and the Chainer ``ResNet50Layers`` it
derives from (chainer/links/model/vision/resnet.py, 4.1.0; topology mirrored by the reference's own
sheep/resnet.py:163-216): conv1 7x7/2 + bias, BN, ReLU, max-pool 3/2 (cover_all), bottleneck stages
[3, 4, 6, 3] with the stride on the first 1x1 conv, ``pool5`` = global average pooling, ``fc6``.

Same attribute names and parameter paths (``res2/a/conv1/W``, ``res3/b2/bn3/gamma`` ...), the same
``functions`` ordered dict and ``__call__(x, layers=[...]) -> dict``.  ``pretrained_model='auto'`` needs a
converted Caffe model that is not available offline; weights are HeNormal(scale=1) like Chainer's
``pretrained_model=None``.  Every residual unit is ONE fused function node
(functions/blocks.py:ResidualUnitFunction); 1x1 convolutions run on the same implicit-GEMM kernels (one tap)."""
import collections

from .. import links as L
from ..functions import blocks
from ..functions import global_average_pooling_2d
from ..runtime.core import Chain


class BottleneckA(Chain):
    def __init__(self, in_channels, mid_channels, out_channels, stride=2, initialW=None):
        super().__init__()
        with self.init_scope():
            self.conv1 = L.Convolution2D(in_channels, mid_channels, 1, stride, 0, initialW=initialW, nobias=True)
            self.bn1 = L.BatchNormalization(mid_channels)
            self.conv2 = L.Convolution2D(mid_channels, mid_channels, 3, 1, 1, initialW=initialW, nobias=True)
            self.bn2 = L.BatchNormalization(mid_channels)
            self.conv3 = L.Convolution2D(mid_channels, out_channels, 1, 1, 0, initialW=initialW, nobias=True)
            self.bn3 = L.BatchNormalization(out_channels)
            self.conv4 = L.Convolution2D(in_channels, out_channels, 1, stride, 0, initialW=initialW, nobias=True)
            self.bn4 = L.BatchNormalization(out_channels)

    def __call__(self, x):
        return blocks.residual_unit(x, [(self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3)],
                                    (self.conv4, self.bn4))


class BottleneckB(Chain):
    def __init__(self, in_channels, mid_channels, initialW=None):
        super().__init__()
        with self.init_scope():
            self.conv1 = L.Convolution2D(in_channels, mid_channels, 1, 1, 0, initialW=initialW, nobias=True)
            self.bn1 = L.BatchNormalization(mid_channels)
            self.conv2 = L.Convolution2D(mid_channels, mid_channels, 3, 1, 1, initialW=initialW, nobias=True)
            self.bn2 = L.BatchNormalization(mid_channels)
            self.conv3 = L.Convolution2D(mid_channels, in_channels, 1, 1, 0, initialW=initialW, nobias=True)
            self.bn3 = L.BatchNormalization(in_channels)

    def __call__(self, x):
        return blocks.residual_unit(x, [(self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3)])


class BuildingBlock(Chain):
    def __init__(self, n_layer, in_channels, mid_channels, out_channels, stride, initialW=None):
        super().__init__()
        with self.init_scope():
            self.a = BottleneckA(in_channels, mid_channels, out_channels, stride, initialW)
            self._forward = ["a"]
            for i in range(n_layer - 1):
                name = 'b{}'.format(i + 1)
                setattr(self, name, BottleneckB(out_channels, mid_channels, initialW))
                self._forward.append(name)

    def __call__(self, x):
        for name in self._forward:
            x = getattr(self, name)(x)
        return x


class ResNet50Layers(Chain):
    def __init__(self, pretrained_model='auto'):
        super().__init__()
        w = L.HeNormal(scale=1.0)
        with self.init_scope():
            self.conv1 = L.Convolution2D(3, 64, 7, 2, 3, initialW=w, dense_rows=True)
            self.bn1 = L.BatchNormalization(64)
            self.res2 = BuildingBlock(3, 64, 64, 256, 1, w)
            self.res3 = BuildingBlock(4, 256, 128, 512, 2, w)
            self.res4 = BuildingBlock(6, 512, 256, 1024, 2, w)
            self.res5 = BuildingBlock(3, 1024, 512, 2048, 2, w)
            self.fc6 = L.Linear(2048, 1000)
        self.pretrained_model = pretrained_model

    exchange_stages = ('res4', 'res5')        # see sheep/resnet.py

    def _stem(self, x):
        return blocks.StemFunction(self.conv1, self.bn1)(x, self.conv1.W, self.conv1.b, self.bn1.gamma, self.bn1.beta)

    @property
    def functions(self):
        # conv1 + bn1 + relu + pool1 are one fused node here; 'conv1' / 'pool1' cannot be tapped separately
        return collections.OrderedDict([
            ('pool1', [self._stem]),
            ('res2', [self.res2]),
            ('res3', [self.res3]),
            ('res4', [self.res4]),
            ('res5', [self.res5]),
            ('pool5', [global_average_pooling_2d]),
            ('fc6', [self._fc6]),
            ('prob', [self._prob]),
        ])

    def _fc6(self, x):
        raise NotImplementedError("the ImageNet classifier head is outside the LoANs training path")

    _prob = _fc6

    @property
    def available_layers(self):
        return list(self.functions.keys())

    def __call__(self, x, layers=['prob']):
        h = x
        activations = {}
        target_layers = set(layers)
        for key, funcs in self.functions.items():
            if len(target_layers) == 0:
                break
            if key in self.exchange_stages:
                h = blocks.stage_boundary(self, key, h)
            for func in funcs:
                h = func(h)
            if key in target_layers:
                activations[key] = h
                target_layers.remove(key)
        return activations


class MyResNet50Layers(ResNet50Layers):

    def __init__(self, *args, **kwargs):
        self.keys_to_remove = kwargs.pop('keys_to_remove', [])
        super().__init__(*args, **kwargs)

    @property
    def functions(self):
        """the parent's layer table without the entries named at construction (reference iou/iou_regressor.py:10-15; a name
        the parent does not have is an error there too)"""
        table = super().functions
        missing = [key for key in self.keys_to_remove if key not in table]
        if missing:
            raise KeyError(missing[0])
        return type(table)((key, funcs) for key, funcs in table.items() if key not in self.keys_to_remove)
