"""``MyResNet50Layers`` (reference iou/iou_regressor.py:4-15) is the ResNet-50 backbone
wrapper of ``Resnet50SheepLocalizer`` (BASELINE config 5).  It is scheduled after the
ResNet-18 path meets its bar; constructing it fails loudly until then."""


class MyResNet50Layers:

    def __init__(self, *args, **kwargs):
        self.keys_to_remove = kwargs.pop('keys_to_remove', [])
        raise NotImplementedError("ResNet-50 backbone (config 5) is not built yet; use SheepLocalizer (--use-resnet-18)")
