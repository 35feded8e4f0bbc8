#!/usr/bin/env python
"""conv1 (dense-row RGB stem, B x 3 x H x W -> 64 channels) forward: the implicit-GEMM tiles against LOANS_TILE_STEM,
interleaved timing.  usage: stem_bench.py [--lib path/to/lib.so] [--batch 256] [--size 224]   (development tool)"""
import argparse
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument('--lib', default='')
ap.add_argument('--batch', type=int, default=256)
ap.add_argument('--size', type=int, default=224)
ap.add_argument('--tiles', default='18,3,10')
args = ap.parse_args()
from loans_amd import _lib  # noqa: E402
if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
from loans_amd import ops  # noqa: E402
B, H = args.batch, args.size
geo = ops.ConvGeometry(B, H, H, 3, 64, 7, 2, 3, dense=True)
x = torch.randn(B, geo.Hp, geo.Wp, 3, device='cuda')
w = torch.randn(64, 7, geo.kwp, 3, device='cuda') * 0.05
bias = torch.randn(64, device='cuda')
y = torch.empty(B, geo.Ho, geo.Wo, 64, device='cuda')
stats = ops.stats_buffer(64, 'cuda')
tiles = [int(t) for t in args.tiles.split(',')]
ts = {t: [] for t in tiles}
for rnd in range(8):
    for t in tiles:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(2):
            ops.conv_fprop(x, w, geo, out=y, bias=bias, stats=stats, tile=t)
        a.record()
        for _ in range(3):
            ops.conv_fprop(x, w, geo, out=y, bias=bias, stats=stats, tile=t)
        b.record()
        torch.cuda.synchronize()
        if rnd:
            ts[t].append(a.elapsed_time(b) / 3)
flop = 2.0 * B * geo.Ho * geo.Wo * 64 * 147
print(' '.join('tile %d: %.3f ms = %.1f TFLOP/s' % (t, np.median(v), flop / np.median(v) / 1e9) for t, v in ts.items()))
