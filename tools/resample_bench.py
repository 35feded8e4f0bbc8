#!/usr/bin/env python
"""Input-contract row: LANCZOS resize + /255 + CHW on the GPU vs Pillow on the host cores.
Prints images/s and the HBM-roofline fraction of the two kernels (algorithmic bytes: frames read once, the uint8
intermediate written and read once, the float32 batch written once)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from PIL import Image
from loans_amd.common.datasets.resample import resize_lanczos

B, H, W, oh, ow = int(os.environ.get('B', 256)), 480, 640, 224, 224
a = np.random.RandomState(0).randint(0, 256, (B, H, W, 3)).astype(np.uint8)
x = torch.from_numpy(a).cuda()
resize_lanczos(x, (oh, ow)); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 20
e0.record()
for _ in range(n):
    y = resize_lanczos(x, (oh, ow))
e1.record(); e1.synchronize()
ms = e0.elapsed_time(e1) / n
byts = B * (H * W * 3 + 2 * H * ow * 3 + oh * ow * 3 * 4)
print('GPU: %d frames %dx%d -> %dx%d: %.3f ms  = %.0f images/s, %.1f GB/s algorithmic = %.1f %% of 8 TB/s'
      % (B, H, W, oh, ow, ms, B / ms * 1e3, byts / ms / 1e6, byts / ms / 1e6 / 8000 * 100))
t0 = time.perf_counter()
m = 32
for b in range(m):
    r = np.asarray(Image.fromarray(a[b]).resize((ow, oh), Image.LANCZOS)).transpose(2, 0, 1).astype(np.float32) / 255
t = (time.perf_counter() - t0) / m
print('host (Pillow, 1 core): %.2f ms per frame = %.0f images/s per core' % (t * 1e3, 1 / t))
h2d = torch.from_numpy(a).pin_memory()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    h2d.to('cuda', non_blocking=True)
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / 5
print('upload of the uint8 frames (pinned): %.2f ms = %.1f GB/s -> %.0f images/s' % (t * 1e3, a.nbytes / t / 1e9, B / t))
