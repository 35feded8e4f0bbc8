#!/usr/bin/env python
"""Race screen of the ping-pong tiles (LOANS_TILE_256x256PP / PP16, csrc/igemm16_pp.h): their LDS-DMA / ds_read ordering rests on
counted vmcnt + raw barriers, and a wrong count shows only when a DMA lands late.  Many launches per shape, bit-compared with the
lock-step tile, while a second stream streams HBM (copies of a 1 GB buffer) and a third runs another convolution, so that the
latencies move.  usage: pp_stress.py [launches per shape]   (development tool)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loans_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ops.set_compute_dtype('bf16'); ops.set_storage_dtype('bf16')
shapes = [(128, 32, 32, 256, 256, 3, 1, 1), (128, 16, 16, 512, 512, 3, 1, 1), (64, 32, 32, 256, 512, 3, 2, 1), (16, 20, 33, 256, 512, 3, 1, 1),
          (64, 32, 32, 1024, 256, 1, 1, 0), (3, 7, 9, 64, 256, 3, 1, 1)]
noise = torch.empty(1 << 28, device='cuda', dtype=torch.float32)
noise2 = torch.empty_like(noise)
s_noise, s_conv = torch.cuda.Stream(), torch.cuda.Stream()
bad = 0
for B, H, W, Cin, Cout, k, s, p in shapes:
    geo = ops.ConvGeometry(B, H, W, Cin, Cout, k, s, p)
    x = torch.randn(B, H, W, Cin, device='cuda').to(torch.bfloat16)
    w16 = ops.cast_bf16(torch.randn(Cout, k, k, Cin, device='cuda') * 0.05)
    ref = ops.conv_fprop(x, w16, geo, tile=9).clone()
    gy = torch.randn(B, geo.Ho, geo.Wo, Cout, device='cuda').to(torch.bfloat16)
    for tile in (43, 44):
        mism = 0
        for it in range(n):
            if it % 4 == 0:
                with torch.cuda.stream(s_noise):
                    noise2.copy_(noise, non_blocking=True)
            if it % 3 == 0:
                with torch.cuda.stream(s_conv):
                    ops.conv_fprop(x, w16, geo, tile=1)
            y = ops.conv_fprop(x, w16, geo, tile=tile)
            if not torch.equal(y, ref):
                mism += 1
        torch.cuda.synchronize()
        bad += mism
        print('shape %s tile %d: %d launches, %d not bit-identical to tile 9' % ((B, H, W, Cin, Cout, k, s, p), tile, n, mism), flush=True)
print('total mismatches: %d' % bad)
sys.exit(1 if bad else 0)
