#!/usr/bin/env python
"""bf16 weight gradients of the stride-1 3x3 layers at configs[2]'s shapes (128 x 3 x 512 x 512): the plain GEMM tiles of
wgrad16_kernel against the halo tiles (csrc/wgrad_halo_bf16.hip), every candidate the autotuner would try, interleaved in one
process.  Development tool.  usage: wgrad_bench.py [--batch 128] [--layers res2,res3] [--reps 5]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get('LOANS_BENCH_LIB'):      # an ablation build of the library (tools/wgrad_ablate.sh)
    from loans_amd import _lib
    _lib.LIB_PATH = os.environ['LOANS_BENCH_LIB']
from loans_amd import ops  # noqa: E402

LAYERS = [('res2', 64, 128, 128, 64), ('res3', 128, 64, 64, 128), ('res4', 256, 32, 32, 256), ('res5', 512, 16, 16, 512),
          ('res6', 512, 8, 8, 512), ('as_r1c0', 128, 37, 37, 128), ('as_r2', 128, 18, 18, 128),
          ('r50_res2', 64, 128, 128, 64), ('r50_res3', 128, 64, 64, 128), ('r50_res4', 256, 32, 32, 256),
          # ResNet-50's 1 x 1 layers (k = 1): a bottleneck's expansion and the next unit's reduction
          ('r50_res2_c3', 64, 128, 128, 256, 1), ('r50_res2_c1', 256, 128, 128, 64, 1), ('r50_res3_c3', 128, 64, 64, 512, 1),
          ('r50_res4_c3', 256, 32, 32, 1024, 1), ('r50_res5_c1', 2048, 16, 16, 512, 1)]


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
ap.add_argument('--batch', type=int, default=128)
ap.add_argument('--layers', default='res2,res3,res4,res5,as_r1c0')
ap.add_argument('--reps', type=int, default=5)
ap.add_argument('--atomics', action='store_true', help='also time every candidate with LOANS_WGRAD_SLABS=0')
args = ap.parse_args()
ops.set_compute_dtype('bf16')
ops.set_storage_dtype('bf16')
for name, Cin, H, W, Cout, *rest in LAYERS:
    if name not in args.layers.split(','):
        continue
    k = rest[0] if rest else 3
    B = args.batch // 2 if name.startswith('r50') else args.batch
    geo = ops.ConvGeometry(B, H, W, Cin, Cout, k, 1, k // 2)
    x = torch.randn((B, H, W, Cin), device='cuda').to(torch.bfloat16)
    gy = torch.randn((B, H, W, Cout), device='cuda').to(torch.bfloat16)
    dw = torch.zeros((Cout, k, k, Cin), device='cuda')
    flop = 2.0 * B * H * W * Cout * k * k * Cin
    alg = (x.numel() + gy.numel()) * 2 + dw.numel() * 4
    plain = ops._wgrad_candidates(geo, ops._WGRAD16_TILES + ((ops.TILE_256x256,) if Cout % 256 == 0 else ()), 32, ops._WGRAD16_TILE_DIMS)
    res = {}
    atomic = {}
    for t in plain + ops._wghalo_candidates(geo):
        ops.WGRAD_SLABS = True
        res[t] = timeit(lambda: ops._conv_wgrad(x, gy, dw, geo, False, 0, t), args.reps)
        if args.atomics:        # the same candidate closing with fp32 atomics into dw (rounds 1-4), interleaved
            ops.WGRAD_SLABS = False
            atomic[t] = timeit(lambda: ops._conv_wgrad(x, gy, dw, geo, False, 0, t), args.reps)
            ops.WGRAD_SLABS = True
    bp = min((t for t in plain), key=res.get)
    halo = [t for t in res if (t & 0xFF) in (38, 39)]
    bh = min(halo, key=res.get) if halo else None
    bound = max(flop / 2.5e15, alg / 6.3e12) * 1e3
    print('%-9s B=%d %dx%dx%d->%d  %.1f GFLOP  bound %.3f ms | plain best tile %d x%d: %.3f ms (%.0f TFLOP/s, %.2f of bound) | halo best %s: %s'
          % (name, B, H, W, Cin, Cout, flop / 1e9, bound, bp & 0xFF, bp >> 8, res[bp], flop / res[bp] / 1e9, bound / res[bp],
             None if bh is None else '%d x%d' % (bh & 0xFF, bh >> 8),
             '' if bh is None else '%.3f ms (%.0f TFLOP/s, %.2f of bound)' % (res[bh], flop / res[bh] / 1e9, bound / res[bh])), flush=True)
    print('      ' + '  '.join('%d/%d:%.3f' % (t & 0xFF, t >> 8, v) for t, v in sorted(res.items(), key=lambda kv: (kv[0] & 0xFF, kv[0] >> 8))), flush=True)
    if atomic:
        ba = min(atomic, key=atomic.get)
        print('      atomics: best %d x%d %.3f ms | ' % (ba & 0xFF, ba >> 8, atomic[ba]) +
              '  '.join('%d/%d:%.3f' % (t & 0xFF, t >> 8, v) for t, v in sorted(atomic.items(), key=lambda kv: (kv[0] & 0xFF, kv[0] >> 8))), flush=True)
