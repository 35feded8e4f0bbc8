"""ResNet-18 variant of the reference (sheep/resnet.py:6-160), MI355X-native.

Same class names, constructor arguments, attribute names and parameter paths
(``conv1``, ``bn1``, ``res2`` ... ``res5``; per block ``conv1/bn1/conv2/bn2``
and the strided 3x3 ``conv3/bn3`` shortcut of ``BasicA``), so snapshots
interchange.  Each block's ``__call__`` is one fused function node
(functions/blocks.py).  Only the basic-block depths the LoANs trainer can
instantiate (18 / 20 / 34) are built; the bottleneck variants
(sheep/resnet.py:163-216) belong to the ResNet-50 localizer.
"""
from .. import links as L
from ..functions import blocks
from ..runtime.core import Chain, ChainList


class ResNet(Chain):

    def __init__(self, n_layers, class_labels=None):
        super(ResNet, self).__init__()
        w = L.HeNormal()
        if n_layers == 18:
            block = [2, 2, 2, 2]
        elif n_layers == 20:
            block = [2, 2, 2, 2, 2, 2]
        elif n_layers == 34:
            block = [3, 4, 6, 3]
        else:
            raise ValueError("You tried to create a ResNet variant that does not exist")
        if class_labels is not None:
            raise NotImplementedError("the ImageNet classification head is outside the LoANs training path")

        with self.init_scope():
            self.conv1 = L.Convolution2D(3, 64, 7, 2, 3, initialW=w, dense_rows=True)
            self.bn1 = L.BatchNormalization(64)
            self.res2 = BasicBlock(block[0], 64, 1, in_ch=64)
            self.res3 = BasicBlock(block[1], 128, in_ch=64)
            self.res4 = BasicBlock(block[2], 256, in_ch=128)
            self.res5 = BasicBlock(block[3], 512, in_ch=256)
            if n_layers == 20:
                self.res6 = BasicBlock(block[4], 512, in_ch=512)
                self.res7 = BasicBlock(block[5], 512, in_ch=512)

        self.n_layers = n_layers
        self.class_labels = class_labels

    # data parallel: the gradient arena is exchanged in three parts, cut in front of these stages -- everything from res5 up
    # (res5 + the head + res6 / res7: 75 % of the bytes at 224 px) as soon as the backward has left res5, res4 (19 %) next,
    # the rest when the backward ends (loans_amd/parallel.py)
    exchange_stages = ('res4', 'res5')

    def __call__(self, x):
        """x: preprocessed frames (``prepare_images``: the padded packed-RGB buffer conv1 reads).  Returns the NHWC feature map."""
        h = blocks.StemFunction(self.conv1, self.bn1)(x, self.conv1.W, self.conv1.b, self.bn1.gamma, self.bn1.beta)
        h = self.res2(h)
        h = self.res3(h)
        h = self.res4(blocks.stage_boundary(self, 'res4', h))
        if hasattr(self, 'res5'):
            h = self.res5(blocks.stage_boundary(self, 'res5', h))
        if hasattr(self, 'res6'):
            h = self.res6(h)
        if hasattr(self, 'res7'):
            h = self.res7(h)
        return h


class BasicBlock(ChainList):

    def __init__(self, layer, ch, stride=2, in_ch=None):
        super(BasicBlock, self).__init__()
        in_ch = ch if in_ch is None else in_ch      # the reference infers it lazily (Convolution2D(None, ...))
        self.add_link(BasicA(ch, stride, in_ch))
        for i in range(layer - 1):
            self.add_link(BasicB(ch))

    def __call__(self, x):
        for f in self.children():
            x = f(x)
        return x


class BasicA(Chain):

    def __init__(self, ch, stride, in_ch=None):
        super(BasicA, self).__init__()
        w = L.HeNormal()
        in_ch = ch if in_ch is None else in_ch
        with self.init_scope():
            self.conv1 = L.Convolution2D(in_ch, ch, 3, stride, 1, initialW=w, nobias=True)
            self.bn1 = L.BatchNormalization(ch)
            self.conv2 = L.Convolution2D(ch, ch, 3, 1, 1, initialW=w, nobias=True)
            self.bn2 = L.BatchNormalization(ch)

            self.conv3 = L.Convolution2D(in_ch, ch, 3, stride, 1, initialW=w, nobias=True)
            self.bn3 = L.BatchNormalization(ch)

    def __call__(self, x):
        return blocks.residual_unit(x, [(self.conv1, self.bn1), (self.conv2, self.bn2)], (self.conv3, self.bn3))


class BasicB(Chain):

    def __init__(self, ch):
        super(BasicB, self).__init__()
        w = L.HeNormal()
        with self.init_scope():
            self.conv1 = L.Convolution2D(ch, ch, 3, 1, 1, initialW=w, nobias=True)
            self.bn1 = L.BatchNormalization(ch)
            self.conv2 = L.Convolution2D(ch, ch, 3, 1, 1, initialW=w, nobias=True)
            self.bn2 = L.BatchNormalization(ch)

    def __call__(self, x):
        return blocks.residual_unit(x, [(self.conv1, self.bn1), (self.conv2, self.bn2)])
