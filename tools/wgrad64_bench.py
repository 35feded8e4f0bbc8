#!/usr/bin/env python
"""bf16 weight gradients of the layers with 64 output channels and 128 < K <= 256 (the dense RGB stem: K = 7 x 24; a ResNet-50
bottleneck's 256 -> 64 reduction): every (tile, slices) candidate the autotuner tries, interleaved.  Development tool.
usage: wgrad64_bench.py [--reps 5]      (LOANS_BENCH_LIB = another build of the library)"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get('LOANS_BENCH_LIB'):
    from loans_amd import _lib
    _lib.LIB_PATH = os.environ['LOANS_BENCH_LIB']
from loans_amd import ops  # noqa: E402


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); fn(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 3)
    return float(np.median(ts))


ap = argparse.ArgumentParser()
ap.add_argument('--reps', type=int, default=5)
args = ap.parse_args()
ops.set_compute_dtype('bf16')
ops.set_storage_dtype('bf16')
cases = [('stem B128 512px', 128, 512, 512, 3, 64, 7, 2, 3, True), ('stem B64 512px', 64, 512, 512, 3, 64, 7, 2, 3, True),
         ('r50 res2 conv1 B64', 64, 128, 128, 256, 64, 1, 1, 0, False)]
for name, B, H, W, Cin, Cout, k, s, p, dense in cases:
    geo = ops.ConvGeometry(B, H, W, Cin, Cout, k, s, p, dense=dense)
    if dense:
        x = ops.prep_images(torch.rand((B, H, W, 3), device='cuda').permute(0, 3, 1, 2).contiguous(), geo)
        dw = torch.zeros((Cout, k, geo.kwp, 3), device='cuda')
    else:
        x = torch.randn((B, H, W, Cin), device='cuda').to(torch.bfloat16)
        dw = torch.zeros((Cout, k, k, Cin), device='cuda')
    gy = torch.randn((B, geo.Ho, geo.Wo, Cout), device='cuda').to(torch.bfloat16)
    alg = (x.numel() + gy.numel()) * 2 + dw.numel() * 4
    cands = ops._wgrad_candidates(geo, ops._WGRAD16_TILES, 32, ops._WGRAD16_TILE_DIMS)
    if dense and ops.stem16_wgrad_ok(geo):
        cands = tuple(cands) + (ops.TILE_STEM,)
    res = {t: timeit(lambda: ops._conv_wgrad(x, gy, dw, geo, False, 0, t), args.reps) for t in cands}
    print('%-20s bytes %.3f GB, at 6.3 TB/s %.3f ms' % (name, alg / 1e9, alg / 6.3e9), flush=True)
    for tile in sorted({t & 0xFF for t in res}):
        row = sorted((t >> 8, v) for t, v in res.items() if (t & 0xFF) == tile)
        print('   tile %2d: ' % tile + '  '.join('x%d:%.3f' % r for r in row), flush=True)
