#!/usr/bin/env python
"""Development tool: latency of SheepLocalizer.predict + assessor score on one frame (the reference's image_sheeping /
BBOXPlotter use), after the tile tables are warm."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import loans_amd  # noqa: E402
from loans_amd.datasets import synthetic  # noqa: E402

hw = int(sys.argv[1]) if len(sys.argv) > 1 else 224
if len(sys.argv) > 2 and sys.argv[2] == 'bf16':
    loans_amd.set_compute_dtype('bf16'); loans_amd.set_storage_dtype('bf16')
np.random.seed(0)
loc, dis = loans_amd.SheepLocalizer((75, 75)), loans_amd.ResnetAssessor()
frames = synthetic.make_frames(3, 1, hw, hw)


def once():
    boxes, rois, _, _ = loc.predict(list(frames))
    with loans_amd.using_config('train', False), loans_amd.using_config('enable_backprop', False):
        return boxes, dis(rois)


for _ in range(3):
    once()
torch.cuda.synchronize()
ts = []
for _ in range(30):
    t0 = time.perf_counter(); once(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print('predict + score, 1 frame %dx%d: median %.2f ms (min %.2f)' % (hw, hw, np.median(ts) * 1e3, min(ts) * 1e3))
