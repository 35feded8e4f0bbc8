#!/usr/bin/env python
"""Per-layer micro-benchmark of the implicit-GEMM kernels (fprop / dgrad / wgrad) on the
ResNet-18-variant and assessor layer shapes, every tile variant interleaved in ONE process
(HIP events, median of `reps`).  Development tool; not part of the product path."""
import argparse
import sys, os
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if '--exp' in sys.argv:      # experiment build (make -C loans_amd/csrc exp): LOANS_DBG bits are read per launch
    from loans_amd import _lib
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libloans_hip_exp.so')
from loans_amd import ops  # noqa: E402

LAYERS = [
    # name, Cin, H, W, Cout, k, stride, pad
    ('stem', 4, 224, 224, 64, 7, 2, 3),
    ('res2', 64, 56, 56, 64, 3, 1, 1),
    ('res3a', 64, 56, 56, 128, 3, 2, 1),
    ('res3', 128, 28, 28, 128, 3, 1, 1),
    ('res4a', 128, 28, 28, 256, 3, 2, 1),
    ('res4', 256, 14, 14, 256, 3, 1, 1),
    ('res5a', 256, 14, 14, 512, 3, 2, 1),
    ('res5', 512, 7, 7, 512, 3, 1, 1),
    ('as_r0c1', 128, 75, 75, 128, 4, 2, 1),
    ('as_r1c0', 128, 37, 37, 128, 3, 1, 1),
    ('as_r2', 128, 18, 18, 128, 3, 1, 1),
]


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--modes', default='fprop,dgrad,wgrad')
    ap.add_argument('--layers', default='')
    ap.add_argument('--tiles', default='0,1,2,3,4,5')
    ap.add_argument('--compute', default='f32')
    ap.add_argument('--inner', type=int, default=3, help='launches per timed sample')
    ap.add_argument('--storage', default='f32', help="bf16: the bf16-storage kernels (loans_igemm_bf16s / loans_wgrad_bf16s); tiles 1,2,3,4,7")
    ap.add_argument('--exp', action='store_true')
    ap.add_argument('--dbg', default='0', help='comma list of LOANS_DBG values to interleave (needs --exp)')
    args = ap.parse_args()
    B = args.batch
    ops.set_compute_dtype(args.compute)
    tiles = [int(t) for t in args.tiles.split(',')]
    names = {0: 'auto', 1: '128x128', 2: '128x64', 3: '64x64', 4: '256x64', 5: '64x128', 6: 'split'}
    names.update({t + 16: n + 'D' for t, n in list(names.items()) if t in (1, 2, 3, 4, 6)})
    names[7] = '256x128'
    names[8] = 'finetail'
    s16 = args.storage == 'bf16'
    if s16:
        ops.set_compute_dtype('bf16')
        ops.set_storage_dtype('bf16')
    for name, Cin, H, W, Cout, k, s, p in LAYERS:
        if args.layers and name not in args.layers.split(','):
            continue
        geo = ops.ConvGeometry(B, H, W, Cin, Cout, k, s, p)
        if s16 and Cin % 8:
            continue
        x = torch.randn(B, H, W, Cin, device='cuda')
        w = torch.randn(Cout, k, k, Cin, device='cuda') * 0.05
        gy = torch.randn(B, geo.Ho, geo.Wo, Cout, device='cuda')
        if s16:
            x, gy = x.to(torch.bfloat16), gy.to(torch.bfloat16)
        y = torch.empty_like(gy)
        gx = torch.empty_like(x)
        dw = torch.zeros_like(w)
        stats = ops.stats_buffer(Cout, 'cuda')
        cin = 3 if Cin == 4 else Cin
        flops = 2.0 * B * geo.Ho * geo.Wo * Cout * k * k * cin
        line = '%-8s M=%8d N=%4d K=%5d |' % (name, B * geo.Ho * geo.Wo, Cout, k * k * Cin)
        cases = []
        for mode in args.modes.split(','):
            for t, dbg in [(t, g) for t in tiles for g in args.dbg.split(',')]:
                if mode == 'wgrad' and t not in (0, 1, 3, 5):
                    continue
                if mode != 'wgrad' and t == 5 or (mode == 'dgrad' and ((t & 15) in (4, 6) or t == 8) and not s16):
                    continue
                if mode == 'fprop':
                    fn = lambda t=t: ops.conv_fprop(x, w, geo, out=y, stats=stats, tile=t)   # noqa: E731
                elif mode == 'dgrad':
                    fn = lambda t=t: ops.conv_dgrad(gy, w, geo, out=gx, tile=t)              # noqa: E731
                else:
                    fn = lambda t=t: ops._conv_wgrad(x, gy, dw, geo, False, 0, t)             # noqa: E731  (sync path: the public op goes to a side stream)
                cases.append((mode, t, dbg, fn, []))
        # every case once per round, rounds repeated: clock / thermal drift hits all cases alike
        for rnd in range(args.reps + 1):
            for mode, t, dbg, fn, ts in cases:
                os.environ['LOANS_DBG'] = dbg
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _ in range(args.inner):
                    fn()
                a.record()
                for _ in range(args.inner):
                    fn()
                b.record()
                torch.cuda.synchronize()
                if rnd:
                    ts.append(a.elapsed_time(b) / args.inner)
        for mode, t, dbg, fn, ts in cases:
            ms = float(np.median(ts))
            line += ' %s/%s%s %6.1f TF' % (mode[0], names.get(t, str(t)), ('#' + dbg) if args.exp else '', flops / ms / 1e9)
        line += ' |'
        print(line, flush=True)


if __name__ == '__main__':
    main()
