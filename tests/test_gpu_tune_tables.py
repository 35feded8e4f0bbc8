"""-m gpu: the kernels bench.py TIMES are the kernels this suite checks.

bench.py runs its four committed workloads on committed tile tables (profiles/<round>_{b256,b128,cfg3,r50}_tune.json: what the timing
autotuner picked in the profiled session -- one tile id per layer shape and operation), while the parity tests run on the fixed
assignment of tests/conftest.py.  This file closes the gap the round-3 verdict named: EVERY entry of the newest committed tables
is launched at its own full shape (B = 256 / 128 / 64) and compared with the plain 128 x 128 implicit-GEMM tile on the same
operands -- the tile the kernel-level tests pin against the oracle on every geometry class."""
import json
import os
import re

import numpy as np
import pytest
import torch

from loans_amd import ops

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _newest_table(name):
    import bench
    return bench._profile_file(name, 'tune')


def _rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-300))


def _operands(geo, s16, gen):
    rnd = lambda *s: torch.randn(*s, device='cuda', generator=gen)        # noqa: E731
    if geo.dense:
        frames = torch.rand(geo.B, 3, geo.H, geo.W, device='cuda', generator=gen)
        frames = (frames * 255).floor() / 255                              # the input contract: uint8 / 255
        x = ops.prep_images(frames, geo)
        w = rnd(geo.Cout, geo.k, geo.kwp, 3) * 0.05 * geo.wmask(x.device)
    else:
        x = rnd(geo.B, geo.H, geo.W, geo.Cin)
        if geo.Cin == 4:
            x[..., 3] = 0
        w = rnd(geo.Cout, geo.k, geo.k, geo.Cin) * (2.0 / (geo.k * geo.k * geo.Cin)) ** 0.5
        if s16:
            x = x.to(torch.bfloat16)
    gy = rnd(geo.B, geo.Ho, geo.Wo, geo.Cout)
    return x, w, (gy.to(torch.bfloat16) if s16 else gy)


def _check_entry(key, mode, tile, gen):
    """returns None if covered and right, a string saying why the entry was not launched otherwise"""
    B, H, W, Cin, Cout, k, stride, pad, dense = key
    s16 = mode.startswith('bf16s_')
    if not (s16 or mode.startswith('f32')):
        return 'compute-only bf16 arm on fp32 tensors (the assessor block that reads the fp32 crops)'
    ops.set_compute_dtype('bf16' if s16 else 'f32')
    ops.set_storage_dtype('bf16' if s16 else 'f32')
    geo = ops.ConvGeometry(B, H, W, Cin, Cout, k, stride, pad, bool(dense))
    x, w, gy = _operands(geo, s16, gen)
    tol_out = 2.0 ** -7 if s16 else 2e-5                   # bf16 outputs: another K order moves a rounding by one spacing
    relu = '_relu' in mode or 'relu_' in mode
    if 'fprop_pair' in mode:
        m = re.search(r'pair(\d+)', mode)
        geo_b = ops.ConvGeometry(B, H, W, Cin, int(m.group(1)) if m else Cout, k, stride, pad)
        wb = torch.randn(geo_b.Cout, k, k, Cin, device='cuda', generator=gen) * (2.0 / (k * k * Cin)) ** 0.5
        sa, sb = ops.stats_buffer(geo.Cout, 'cuda'), ops.stats_buffer(geo_b.Cout, 'cuda')
        ya, yb = ops.conv_fprop_pair(x, w, wb, geo, geo_b, sa, sb, tile=tile)
        ta, tb = ops.stats_buffer(geo.Cout, 'cuda'), ops.stats_buffer(geo_b.Cout, 'cuda')
        ra, rb = ops.conv_fprop(x, w, geo, stats=ta, tile=1), ops.conv_fprop(x, wb, geo_b, stats=tb, tile=1)
        assert _rel(ya, ra) <= tol_out and _rel(yb, rb) <= tol_out, (key, mode, hex(tile))
        # statistics come from the fp32 accumulators on both sides (sum, sum of squares per channel)
        assert _l2(sa.sum(0), ta.sum(0)) < 1e-5 and _l2(sb.sum(0), tb.sum(0)) < 1e-5, (key, mode, hex(tile))
    elif 'fprop' in mode:
        st = ops.stats_buffer(geo.Cout, 'cuda') if '_stats' in mode else None
        y = ops.conv_fprop(x, w, geo, stats=st, relu_in=relu, tile=tile)
        st1 = ops.stats_buffer(geo.Cout, 'cuda') if st is not None else None
        ref = ops.conv_fprop(x, w, geo, stats=st1, relu_in=relu, tile=1)
        assert _rel(y, ref) <= tol_out and _l2(y, ref) < (1e-3 if s16 else 1e-5), (key, mode, hex(tile), _rel(y, ref))
        if st is not None:                                 # statistics of the fp32 accumulators on both sides
            assert _l2(st.sum(0), st1.sum(0)) < 1e-5, (key, mode, hex(tile), _l2(st.sum(0), st1.sum(0)))
    elif 'dgrad' in mode:
        if not geo.dgrad:
            return 'no data gradient for this geometry'
        if mode.endswith('_bn'):
            C_ = geo.Cin
            y = torch.randn(B, H, W, C_, device='cuda', generator=gen) * 1.5
            y = y.to(torch.bfloat16) if s16 else y
            stats = ops.stats_buffer(C_, 'cuda')
            flat = y.double().reshape(-1, C_)
            stats[0, 0], stats[0, 1] = flat.sum(0), (flat * flat).sum(0)
            one = torch.ones(C_, device='cuda')
            bst = ops.bn_finalize(stats, B * H * W, one * 1.1, one * 0.2, torch.zeros(C_, device='cuda'), torch.ones(C_, device='cuda'))
            gx, sums = ops.conv_dgrad(gy, w, geo, tile=tile, bn_sums=(y, bst))
            ref, rsums = ops.conv_dgrad(gy, w, geo, tile=1, bn_sums=(y, bst))
            assert _l2(sums.sum(0), rsums.sum(0)) < (2e-2 if s16 else 1e-4), (key, mode, hex(tile))
        else:
            gx = ops.conv_dgrad(gy, w, geo, tile=tile)
            ref = ops.conv_dgrad(gy, w, geo, tile=1)
        assert _rel(gx, ref) <= tol_out and _l2(gx, ref) < (1e-3 if s16 else 1e-5), (key, mode, hex(tile), _rel(gx, ref))
    elif 'wgrad' in mode:
        dw, ref = torch.zeros(geo.w_numel, device='cuda'), torch.zeros(geo.w_numel, device='cuda')
        ops._conv_wgrad(x, gy, dw, geo, relu, 0, tile)
        ops._conv_wgrad(x, gy, ref, geo, relu, 0, 1)
        assert _rel(dw, ref) <= 1e-4, (key, mode, hex(tile), _rel(dw, ref))
    else:
        return 'unknown mode'
    return None


@pytest.mark.parametrize("name", ['b256', 'b128', 'cfg3', 'r50'])        # b128: one rank's shard of configs[3] (round 6)
def test_every_entry_of_the_committed_tile_table_against_the_plain_tile(name):
    path = _newest_table(name)
    assert path and os.path.exists(path), 'no committed tile table for %s' % name
    doc = json.load(open(path))
    assert doc['stamp']['arch'] == 'gfx950' and doc['stamp']['schema'] == ops.TUNE_SCHEMA
    gen = torch.Generator(device='cuda').manual_seed(5)
    ran, skipped = 0, {}
    old = (ops.COMPUTE, ops.STORAGE)
    try:
        for ks, modes in sorted(doc['entries'].items()):
            key = tuple(int(v) for v in ks.split(','))
            for mode, tile in sorted(modes.items()):
                why = _check_entry(key, mode, int(tile), gen)
                if why is None:
                    ran += 1
                else:
                    skipped[why] = skipped.get(why, 0) + 1
                torch.cuda.synchronize()
    finally:
        ops.set_compute_dtype(old[0])
        ops.set_storage_dtype(old[1])
    print('%s: %d table entries launched at their own shape and compared, not covered: %s' % (os.path.basename(path), ran, skipped))
    assert ran >= 0.85 * (ran + sum(skipped.values())) and ran >= 30
