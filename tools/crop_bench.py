#!/usr/bin/env python
"""Standalone timing of the crop-gradient kernels at the bench shape (B x 75 x 75 crops, 128 channels):
loans_crop_dgrad (one launch) against the per-class loans_dgrad_c4 launches it replaces.  usage: crop_bench.py [B] [bf16]"""
import sys
import numpy as np
import torch
import os
sys.path.insert(0, '.')
from loans_amd import _lib, ops
if os.environ.get('LOANS_CROP_DBG'):        # experiment build (make -C loans_amd/csrc exp)
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libloans_hip_exp.so')

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g16 = len(sys.argv) > 2 and sys.argv[2] == 'bf16'
H = W = 75
Cc = 128
ga, gb = ops.ConvGeometry(B, H, W, 4, Cc, 3, 1, 1), ops.ConvGeometry(B, H, W, 4, Cc, 4, 2, 1)
dt = torch.bfloat16 if g16 else torch.float32
gya = torch.randn(B, ga.Ho, ga.Wo, Cc, device='cuda').to(dt)
gyb = torch.randn(B, gb.Ho, gb.Wo, Cc, device='cuda').to(dt)
wa = torch.randn(Cc, 3, 3, 4, device='cuda') * 0.05
wb = torch.randn(Cc, 4, 4, 4, device='cuda') * 0.05


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def old():
    ops.CROP_DGRAD = False
    ref = ops.conv_dgrad(gyb, wb, gb)
    ops.conv_dgrad(gya, wa, ga, out=ref, addend=ref)
    ops.CROP_DGRAD = True
    return ref


new = lambda: ops.crop_dgrad(gya, wa, ga, gyb, wb, gb)      # noqa: E731
t_new, t_old = timeit(new), timeit(old)
nbytes = (gya.numel() + gyb.numel()) * gya.element_size() + B * H * W * 16
print('B=%d %s: crop_dgrad %.3f ms (%.0f GB/s of algorithmic bytes %.0f MB)   per-class kernels %.3f ms   max|diff| %.2e'
      % (B, 'bf16' if g16 else 'fp32', t_new, nbytes / t_new / 1e6, nbytes / 1e6, t_old, float((new() - old()).abs().max())))
