#!/usr/bin/env python
"""Where the HOST's time goes while it enqueues a step (cProfile over a few update_core calls): the Python / ctypes / torch
calls behind ~1000 launches.  Tensor sizes barely matter for this, so it runs on small frames.
usage: host_profile.py [--resnet50] [--dtype bf16] [B HW]"""
import cProfile
import os
import pstats
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import loans_amd                                   # noqa: E402
from loans_amd import ops                          # noqa: E402
from loans_amd.datasets import synthetic           # noqa: E402
from loans_amd.runtime import training             # noqa: E402

r50 = '--resnet50' in sys.argv
bf16 = '--dtype' in sys.argv and sys.argv[sys.argv.index('--dtype') + 1] == 'bf16'
nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
B, hw = (nums + [4, 320])[:2] if len(nums) < 2 else nums[:2]
crop = 75
if bf16:
    ops.set_compute_dtype('bf16'); ops.set_storage_dtype('bf16')
dev = torch.device('cuda', 0)
frames = torch.from_numpy(synthetic.make_frames(1, B, hw, hw)).to(dev)
real, labels = synthetic.make_assessor_batch(2, B, crop, crop)
real, labels = torch.from_numpy(real).to(dev), torch.from_numpy(labels).to(dev)
np.random.seed(0)
loc = (loans_amd.Resnet50SheepLocalizer if r50 else loans_amd.SheepLocalizer)((crop, crop))
loc.param_predictor.W.set_logical((1e-3 * np.random.standard_normal(loc.param_predictor.W.logical_shape)).astype(np.float32))
dis = loans_amd.ResnetAssessor()
with loans_amd.using_config('enable_backprop', False):
    dis(real[:2])
loc.finalize(dev)
upd = loans_amd.SheepAssessor(models=[loc, dis], iterator={'main': training.DeviceBatchIterator([frames]),
                                                            'real': training.DeviceBatchIterator([(real, labels)])},
                              optimizer={'opt_gen': loans_amd.Adam(alpha=1e-3, amsgrad=True).setup(loc),
                                         'opt_dis': loans_amd.Adam(alpha=1e-3, amsgrad=True).setup(dis)},
                              converter=training.identity_converter, device=0)
for _ in range(3):
    upd.update()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(5):
    torch.cuda.synchronize()
    upd.update()
print('host enqueue per step (GPU idle at the start of each): %.2f ms' % ((time.perf_counter() - t0) / 5 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    torch.cuda.synchronize()
    upd.update()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
st.sort_stats('cumulative').print_stats(22)
