#!/usr/bin/env python
"""Development probe: the Adam-AMSGrad update and the arena cast ALONE (nothing else on the device), operands cold, at the
ResNet-50 localizer's and the assessor's parameter counts -- against what they take at the end of a step (profiles/r4_r50_trace_summary.txt:
2.2 / 1.7 TB/s)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                # noqa: E402
if os.environ.get('LOANS_BENCH_LIB'):      # another build of the library
    from loans_amd import _lib
    _lib.LIB_PATH = os.environ['LOANS_BENCH_LIB']
from loans_amd import ops   # noqa: E402

scrub = torch.empty(512 << 20, device='cuda', dtype=torch.uint8)


def timed(fn, reps=6):
    fn()
    best = 1e9
    for _ in range(reps):
        scrub.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def heat(ms=28.0):
    """~ms of back-to-back MFMA work (a res3-sized 3x3 bf16 convolution), like the step the update follows"""
    geo = ops.ConvGeometry(64, 64, 64, 128, 128, 3, 1, 1)
    if not hasattr(heat, 'x'):
        ops.set_compute_dtype('bf16'); ops.set_storage_dtype('bf16')
        heat.x = torch.randn((64, 64, 64, 128), device='cuda').to(torch.bfloat16)
        heat.w = torch.randn((128, 3, 3, 128), device='cuda') * 0.03
        heat.out = torch.empty((64, 64, 64, 128), device='cuda', dtype=torch.bfloat16)
        heat.one = timed(lambda: ops.conv_fprop(heat.x, heat.w, geo, out=heat.out, tile=1), reps=3)
    for _ in range(max(1, int(ms / heat.one))):
        ops.conv_fprop(heat.x, heat.w, geo, out=heat.out, tile=1)


def timed_hot(fn, reps=6):
    fn()
    best, worst = 1e9, 0.0
    for _ in range(reps):
        heat()
        scrub.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        t = e0.elapsed_time(e1)
        best, worst = min(best, t), max(worst, t)
    return best, worst


for n in (25_600_000, 36_000_000, 48_000_000, 78_000_000):
    p, g, m, v, vh = (torch.rand(n, device='cuda') * 0.01 for _ in range(5))
    ms = timed(lambda: ops.adam_amsgrad(p, g, m, v, vh, 1e-4, 0.9, 0.999, 1e-8, 1.0, 0.0))
    print('adam_amsgrad n = %d: %.3f ms = %.2f TB/s (36 bytes per parameter)' % (n, ms, n * 36 / ms * 1e-9), flush=True)
    b, w_ = timed_hot(lambda: ops.adam_amsgrad(p, g, m, v, vh, 1e-4, 0.9, 0.999, 1e-8, 1.0, 0.0))
    print('   right behind 28 ms of MFMA work: %.3f - %.3f ms' % (b, w_), flush=True)
    ms = timed(lambda: ops.cast_bf16(p))
    print('cast_bf16    n = %d: %.3f ms = %.2f TB/s (6 bytes per parameter)' % (n, ms, n * 6 / ms * 1e-9), flush=True)
