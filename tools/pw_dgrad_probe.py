#!/usr/bin/env python
"""Development probe: the data gradients of ResNet-50's 1x1 reductions (the GEMM shape of the expansions' forward: K = 64 / 128 / 256,
N = 4 K) with the identity-shortcut epilogue (addend masked by the block output's ReLU), every tile on offer, cold and warm,
against the HBM bound of their four tensors."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                # noqa: E402
from loans_amd import ops   # noqa: E402

ops.set_compute_dtype('bf16'); ops.set_storage_dtype('bf16')
B = 64
scrub = torch.empty(512 << 20, device='cuda', dtype=torch.uint8)


def timed(fn, cold, reps=6):
    fn()
    best = 1e9
    for _ in range(reps):
        if cold:
            scrub.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


for name, Cin, HW, Cout in [('res2 reduce', 256, 128, 64), ('res3 reduce', 512, 64, 128), ('res4 reduce', 1024, 32, 256)]:
    geo = ops.ConvGeometry(B, HW, HW, Cin, Cout, 1, 1, 0)
    w = torch.randn((Cout, 1, 1, Cin), device='cuda') * 0.05
    gy = torch.randn((B, HW, HW, Cout), device='cuda').to(torch.bfloat16)
    add = torch.randn((B, HW, HW, Cin), device='cuda').to(torch.bfloat16)
    ref = torch.randn((B, HW, HW, Cin), device='cuda').to(torch.bfloat16)
    out = torch.empty((B, HW, HW, Cin), device='cuda', dtype=torch.bfloat16)
    mb = B * HW * HW * (Cout + 3 * Cin) * 2 / 1e6
    row = []
    for tile in (1, 2, 7, 9, 11, 36):
        try:
            for cold in (True, False):
                ms = timed(lambda: ops.conv_dgrad(gy, w, geo, out=out, addend=add, addend_mask_ref=ref, tile=tile), cold)
                row.append('%d %s %.3f' % (tile, 'cold' if cold else 'warm', ms))
        except Exception as e:
            row.append('%d: %s' % (tile, type(e).__name__))
    print('%s dgrad + masked addend: %.0f MB = %.3f ms at 6.3 TB/s | %s' % (name, mb, mb / 6.3e3 / 1e3, ' | '.join(row)), flush=True)
