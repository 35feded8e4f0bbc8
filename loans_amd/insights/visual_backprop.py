"""``VisualBackprop`` (reference insights/visual_backprop.py:10-53): a saliency map of the localizer's decision -- start from
the channel mean of the last feature map and walk the network back along the FIRST input of every node; at each convolution
and pooling node, upsample the map to the node's input size with an all-ones transposed convolution of the node's own
kernel footprint / stride / padding, and multiply by the channel mean of that input; finally min-max normalise per image.

The reference walks Chainer's per-op graph.  Here a residual unit is one fused node, so the forward pass leaves the walk's
raw material behind instead: with ``ops.VBP_TAPS`` set to a list, every main-branch convolution and the stem's pooling append
``(channel mean of their input, kernel, stride, pad)`` in forward order (functions/blocks.py, SheepLocalizer.__call__), and
the anchor variable carries that list.  The shortcut branches are never visited -- as in the reference, where the residual
sum's first input is the main branch (sheep/resnet.py:137-141,157-160)."""
import torch

from .. import ops
from ..runtime.core import Variable, no_backprop_mode


class VisualBackprop:

    def __init__(self):
        self.xp = None

    def scale_layer(self, feature_map, tap):
        """visual_backprop.py:26-41 for one recorded node"""
        return ops.vbp_scale(feature_map, tap['avg'], tap['k'], tap['s'], tap['p'])

    def perform_visual_backprop(self, variable):
        taps = getattr(variable, 'vbp_taps', None)
        if taps is None:
            raise RuntimeError('run the forward pass with ops.VBP_TAPS = [] (SheepLocalizer.predict(..., '
                               'return_visual_backprop=True) does): the fused blocks record what the walk needs')
        data = variable.data if isinstance(variable, Variable) else variable
        self.xp = torch
        with no_backprop_mode():
            visualization = ops.channel_mean(data)                    # F.average(variable, axis=1, keepdims=True)
            for tap in reversed(taps):
                visualization = self.scale_layer(visualization, tap)
            ops.minmax_normalize_(visualization)
        return visualization.unsqueeze(1)                             # (B, 1, H, W) like the reference's
