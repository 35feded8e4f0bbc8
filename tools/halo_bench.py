#!/usr/bin/env python
"""One bf16-storage convolution at a given tile, timed alone: halo_bench.py B H W Cin Cout tile[,tile...]  (LOANS_HALO_DBG with
the experiment build: ablation bits of the weight-stationary kernel: 1 no image loads, 2 no MFMAs, 4 no epilogue, 8 no stores)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loans_amd import _lib, ops
if os.environ.get('LOANS_HALO_DBG'):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libloans_hip_exp.so')
if os.environ.get('HALO_BENCH_LIB'):            # (this tool's own variable: another build of the library, for A/B runs)
    _lib.LIB_PATH = os.path.abspath(os.environ['HALO_BENCH_LIB'])
B, H, W, Cin, Cout = (int(v) for v in sys.argv[1:6])
tiles = [int(t) for t in sys.argv[6].split(',')]
ops.set_compute_dtype('bf16'); ops.set_storage_dtype('bf16')
geo = ops.ConvGeometry(B, H, W, Cin, Cout, 3, 1, 1)
x = torch.randn(B, H, W, Cin, device='cuda').to(torch.bfloat16)
w = (torch.randn(Cout, 3, 3, Cin, device='cuda') * 0.05)
w16 = ops.cast_bf16(w)
first = None
for t in tiles:
    stats = ops.stats_buffer(Cout, 'cuda')
    fn = lambda: ops.conv_fprop(x, w16, geo, stats=stats, tile=t)
    y = fn(); torch.cuda.synchronize()
    if first is None:
        first = y.clone()
    else:
        print('tile %3d vs tile %3d: max |diff| %.3g, identical %s' % (t, tiles[0], float((y.float() - first.float()).abs().max()), torch.equal(y, first)))
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts))
    fl = 2 * B * H * W * Cout * 9 * Cin
    print('tile %3d: %.3f ms  %.0f TFLOP/s  %.0f GB/s (in + out)' % (t, ms, fl / ms / 1e9, (x.numel() + B * H * W * Cout) * 2 / ms / 1e6))
