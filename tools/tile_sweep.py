#!/usr/bin/env python
"""Every tile form the selection may pick for a layer shape, against the plain 128 x 128 tile on the same operands: forward,
data gradient (plain / + addend in place / + masked addend / masked), weight gradient.  The parity session assigns tiles by a
hash of the shape (ops.TUNE_POLICY = 'fixed'), the timing autotuner by speed: either way any candidate may end up in a step, so
each must be right on every shape it is offered for.  usage: tile_sweep.py [B H W] [--bf16]   (default: the ResNet-18 stages
of a 3 x 320 x 304 step, the shapes of tests/test_gpu_tall_frames.py)"""
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from loans_amd import ops          # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith('--')]
B, H, W = (int(v) for v in args[:3]) if len(args) >= 3 else (3, 320, 304)
h, w = ops.conv_outsize(ops.conv_outsize(H, 7, 2, 3), 3, 2, 0, True), ops.conv_outsize(ops.conv_outsize(W, 7, 2, 3), 3, 2, 0, True)
shapes, c = [(h, w, 64, 64, 1)], 64
for cout in (128, 256, 512, 512, 512):
    if min(h, w) < 2:
        break
    shapes.append((h, w, c, cout, 2))
    h, w, c = ops.conv_outsize(h, 3, 2, 1), ops.conv_outsize(w, 3, 2, 1), cout
    shapes.append((h, w, c, c, 1))
gen = torch.Generator(device='cuda').manual_seed(0)
rnd = lambda *s: torch.randn(*s, device='cuda', generator=gen)      # noqa: E731
worst = 0.0


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def report(what, geo, tile, e, tol=2e-5):
    global worst
    worst = max(worst, e)
    flag = '' if e <= tol else '   <-- WRONG'
    if flag or os.environ.get('SWEEP_VERBOSE'):
        print('%-34s %3dx%3dx%3d -> %3d s%d  tile %-8s rel err %.2e%s' % (what, geo.H, geo.W, geo.Cin, geo.Cout, geo.stride, hex(tile), e, flag), flush=True)


for (h, w, cin, cout, s) in shapes:
    geo = ops.ConvGeometry(B, h, w, cin, cout, 3, s, 1)
    x, wt = rnd(B, h, w, cin), rnd(cout, 3, 3, cin) * 0.05
    gy = rnd(B, geo.Ho, geo.Wo, cout)
    M, nch = B * geo.Ho * geo.Wo, (9 * cin + 31) // 32
    # ---- forward ----
    ref = ops.conv_fprop(x, wt, geo, tile=1)
    cands = set(ops._FPROP_TILES) | set(ops._splitk_candidates(M, cout, nch))
    if ops._finetail_plan(M, cout, nch, x.device)[1] > 1:
        cands |= {ops.TILE_FINETAIL, ops.TILE_FINETAIL | 16}
    for t in sorted(cands):
        st = ops.stats_buffer(cout, 'cuda')
        y = ops.conv_fprop(x, wt, geo, stats=st, tile=t)
        report('fprop', geo, t, rel(y, ref))
        report('fprop statistics', geo, t, rel(st.sum(0)[0].float(), ref.double().sum((0, 1, 2)).float()), 1e-4)
    # ---- data gradient, the four epilogues the residual units use ----
    cls_rows = B * geo.dgrad[0][0].gridH * geo.dgrad[0][0].gridW
    sk = ops._splitk_candidates(cls_rows, cin, (min(d.ntaps for d, _, _ in geo.dgrad) * cout + 31) // 32)
    cands = sorted(set(ops._IGEMM_TILES) | set(ops._class_candidates(geo)) | set(sk))
    other, keep = rnd(B, h, w, cin), rnd(B, h, w, cin)
    for name, kw in (('dgrad', {}), ('dgrad * mask', dict(mask_ref=keep)),
                     ('dgrad + masked addend', dict(addend=other, addend_mask_ref=keep)), ('dgrad + addend in place', 'inplace'),
                     ('dgrad * mask + addend in place', 'inplace_mask')):
        def run(t):
            if kw == 'inplace':
                out = other.clone()
                return ops.conv_dgrad(gy, wt, geo, out=out, addend=out, tile=t)
            if kw == 'inplace_mask':
                out = other.clone()
                return ops.conv_dgrad(gy, wt, geo, out=out, mask_ref=keep, addend=out, tile=t)
            return ops.conv_dgrad(gy, wt, geo, tile=t, **kw)
        ref = run(1)
        for t in cands:
            if kw == 'inplace_mask' and (t >> 8) and not (t & ops.TILE_CLASSES):
                continue                      # split-K cannot sum a masked term into an aliased addend (ops.conv_dgrad asserts)
            report(name, geo, t, rel(run(t), ref))
    # ---- weight gradient ----
    def wg(t):
        dw = torch.zeros(cout, 3, 3, cin, device='cuda')
        ops._conv_wgrad(x, gy, dw, geo, False, 0, t)
        return dw
    ref = wg(1)
    for t in ops._wgrad_candidates(geo, ops._WGRAD_TILES, 32):
        report('wgrad', geo, t, rel(wg(t), ref), 5e-5)
torch.cuda.synchronize()
print('tile sweep over %d layer shapes of a %d x %d x %d step: worst relative difference %.2e' % (len(shapes), B, H, W, worst))
