#!/usr/bin/env python
"""Development probe: what the BN-statistics epilogue (LOANS_F_STATS) and the BN-sums epilogue (LOANS_F_BNSUMS) add to the bf16
convolutions of configs[2], solo (same tile with and without the flag)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loans_amd import ops  # noqa: E402

ops.set_compute_dtype('bf16'); ops.set_storage_dtype('bf16')
LAYERS = [('res2', 64, 128, 128, 64), ('res3', 128, 64, 64, 128), ('res4', 256, 32, 32, 256), ('res5', 512, 16, 16, 512)]
B = 128


def t(fn, reps=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 2)
    return float(np.median(ts))


for name, Cin, H, W, Cout in LAYERS:
    geo = ops.ConvGeometry(B, H, W, Cin, Cout, 3, 1, 1)
    x = torch.randn((B, H, W, Cin), device='cuda').to(torch.bfloat16)
    w = torch.randn((Cout, 3, 3, Cin), device='cuda') * 0.05
    y = ops.conv_fprop(x, w, geo)                       # tunes 'bf16s_fprop...'
    stats = ops.stats_buffer(Cout, x.device)
    ops.conv_fprop(x, w, geo, stats=stats)              # tunes the stats key
    tile = geo.tuned[[k for k in geo.tuned if 'fprop_stats' in k][0]]
    a = t(lambda: ops.conv_fprop(x, w, geo, tile=tile))
    b = t(lambda: ops.conv_fprop(x, w, geo, stats=stats, tile=tile))
    gy = torch.randn((B, H, W, Cout), device='cuda').to(torch.bfloat16)
    st = ops.BNState(Cin, x.device)
    for v in (st.mean, st.rstd, st.scale, st.shift):
        v.fill_(0.5)
    ops.conv_dgrad(gy, w, geo)
    ops.conv_dgrad(gy, w, geo, bn_sums=(x, st))
    dt = geo.tuned[[k for k in geo.tuned if k.endswith('_bn')][0]]
    c = t(lambda: ops.conv_dgrad(gy, w, geo, tile=dt))
    d = t(lambda: ops.conv_dgrad(gy, w, geo, tile=dt, bn_sums=(x, st)))
    print('%-5s fprop tile %3d: %.3f ms, + statistics %.3f ms (+%.0f us) | dgrad tile %3d: %.3f ms, + BN sums %.3f ms (+%.0f us)'
          % (name, tile, a, b, (b - a) * 1e3, dt, c, d, (d - c) * 1e3), flush=True)
