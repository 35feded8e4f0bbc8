#!/usr/bin/env python
"""Where the MFMA time of one joint step goes: every distinct conv of the step (with its multiplicity in
fprop / dgrad / wgrad) timed solo with the autotuned tile; prints time per step and achieved TFLOP/s."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loans_amd import ops
from tools.conv_bench import timeit

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
# name, Cin, H, W, Cout, k, s, p, n_fprop, n_dgrad, n_wgrad   (localizer at 224^2; assessor at 75^2: 2 fwd, A: dgrad only, B: wgrad + dgrad)
L = [
    ('stem',     4, 224, 224,  64, 7, 2, 3, 1, 0, 1),
    ('res2',    64,  56,  56,  64, 3, 1, 1, 5, 5, 5),
    ('res3a',   64,  56,  56, 128, 3, 2, 1, 2, 2, 2),
    ('res3',   128,  28,  28, 128, 3, 1, 1, 3, 3, 3),
    ('res4a',  128,  28,  28, 256, 3, 2, 1, 2, 2, 2),
    ('res4',   256,  14,  14, 256, 3, 1, 1, 3, 3, 3),
    ('res5a',  256,  14,  14, 512, 3, 2, 1, 2, 2, 2),
    ('res5',   512,   7,   7, 512, 3, 1, 1, 3, 3, 3),
    ('a_r0c0',   4,  75,  75, 128, 3, 1, 1, 2, 1, 1),
    ('a_r0cs',   4,  75,  75, 128, 4, 2, 1, 2, 1, 1),
    ('a_r0c1', 128,  75,  75, 128, 4, 2, 1, 2, 2, 1),
    ('a_r1c0', 128,  37,  37, 128, 3, 1, 1, 2, 2, 1),
    ('a_r1c1', 128,  37,  37, 128, 4, 2, 1, 4, 4, 2),
    ('a_r23',  128,  18,  18, 128, 3, 1, 1, 8, 8, 4),
]
tot = {'fprop': 0.0, 'dgrad': 0.0, 'wgrad': 0.0}
totf = {'fprop': 0.0, 'dgrad': 0.0, 'wgrad': 0.0}
print('%-8s | %-22s | %-22s | %-22s' % ('layer', 'fprop  n x ms (TF)', 'dgrad  n x ms (TF)', 'wgrad  n x ms (TF)'))
for name, Cin, H, W, Cout, k, s, p, nf, nd, nw in L:
    if name == 'stem':          # dense K rows on the padded packed-RGB frame buffer (LOANS_F_DENSE)
        geo = ops.ConvGeometry(B, H, W, 3, Cout, k, s, p, dense=True)
        x = torch.randn(B, geo.Hp, geo.Wp, 3, device='cuda'); w = torch.randn(Cout, k, geo.kwp, 3, device='cuda') * 0.05
    else:
        geo = ops.ConvGeometry(B, H, W, Cin, Cout, k, s, p)
        x = torch.randn(B, H, W, Cin, device='cuda'); w = torch.randn(Cout, k, k, Cin, device='cuda') * 0.05
    gy = torch.randn(B, geo.Ho, geo.Wo, Cout, device='cuda'); y = torch.empty_like(gy); gx = torch.empty(B, H, W, Cin, device='cuda'); dw = torch.zeros_like(w)
    stats = ops.stats_buffer(Cout, 'cuda') if name[0] != 'a' else None
    cin = 3 if Cin == 4 else Cin
    flops = 2.0 * B * geo.Ho * geo.Wo * Cout * k * k * cin
    cells = []
    for mode, n, fn in (('fprop', nf, lambda: ops.conv_fprop(x, w, geo, out=y, stats=stats)),
                        ('dgrad', nd, lambda: ops.conv_dgrad(gy, w, geo, out=gx)),
                        ('wgrad', nw, lambda: ops._conv_wgrad(x, gy, dw, geo, False, 0, 0))):
        if n == 0:
            cells.append('-'); continue
        fn(); ms = timeit(fn, 5)
        tot[mode] += n * ms; totf[mode] += n * flops
        cells.append('%d x %.3f (%5.1f)' % (n, ms, flops / ms / 1e9))
    print('%-8s | %-22s | %-22s | %-22s' % (name, *cells), flush=True)
for m in tot:
    print('%s: %.2f ms per step, %.1f TFLOP/s' % (m, tot[m], totf[m] / tot[m] / 1e9))
print('all convs: %.2f ms, %.1f TFLOP/s' % (sum(tot.values()), sum(totf.values()) / sum(tot.values()) / 1e9))
