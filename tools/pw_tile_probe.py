#!/usr/bin/env python
"""Development probe: LOANS_TILE_PW against the implicit-GEMM tiles on ResNet-50's res2 / res3 expansions (B = 64 of 512 x 512
frames), with and without BN statistics, operands cold (a 512 MB fill before every launch) and warm."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                # noqa: E402
from loans_amd import ops   # noqa: E402

ops.set_compute_dtype('bf16'); ops.set_storage_dtype('bf16')
B = 64
scrub = torch.empty(512 << 20, device='cuda', dtype=torch.uint8)


def timed(fn, cold, reps=8):
    fn()
    best = 1e9
    for _ in range(reps):
        if cold:
            scrub.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


for name, Cin, HW, Cout in [('res2 expand', 64, 128, 256), ('res3 expand', 128, 64, 512), ('res4 expand', 256, 32, 1024)]:
    geo = ops.ConvGeometry(B, HW, HW, Cin, Cout, 1, 1, 0)
    x = torch.randn((B, HW, HW, Cin), device='cuda').to(torch.bfloat16)
    w = (torch.randn((Cout, 1, 1, Cin), device='cuda') * 0.05).to(torch.bfloat16)
    out = torch.empty((B, HW, HW, Cout), device='cuda', dtype=torch.bfloat16)
    mb = B * HW * HW * (Cin + Cout) * 2 / 1e6
    for stats in (True, False):
        st = ops.stats_buffer(Cout, x.device) if stats else None
        row = []
        for tile in (9, 36, ops.TILE_PW):
            for cold in (True, False):
                ms = timed(lambda: ops.conv_fprop(x, w, geo, out=out, stats=st, tile=tile), cold)
                row.append('%s %s %.3f ms (%.2f TB/s)' % (tile, 'cold' if cold else 'warm', ms, mb / ms * 1e-6))
        print('%s stats=%d: %s' % (name, stats, ' | '.join(row)), flush=True)
