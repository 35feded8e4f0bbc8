"""Outputs of the kernels whose stores take a cache policy (convolution epilogues of every tile family, the stem, the BN
apply passes), on fixed seeded inputs.  tests/test_gpu_nontemporal.py runs this module twice -- in its own process with the
default policy and in a child whose environment forces the non-temporal form -- and compares the bytes."""
import sys

import numpy as np
import torch


def outputs():
    from loans_amd import ops
    dev = torch.device('cuda', 0)
    g = torch.Generator(device='cpu').manual_seed(7)
    rnd = lambda *s: torch.randn(s, generator=g)      # noqa: E731
    out = {}
    old = (ops.COMPUTE, ops.STORAGE)
    try:
        for dtype in ('f32', 'bf16'):
            _one_arm(ops, dev, rnd, dtype, out)
    finally:                            # the arm is process-wide state: later tests of the same run expect it back
        ops.set_compute_dtype(old[0])
        ops.set_storage_dtype(old[1])
    return out


def _one_arm(ops, dev, rnd, dtype, out):
    ops.set_compute_dtype(dtype)
    ops.set_storage_dtype(dtype)
    cast = (lambda t: t.to(dev).to(torch.bfloat16)) if dtype == 'bf16' else (lambda t: t.to(dev))
    # (name, B, H, W, Cin, Cout, k, stride, pad, tiles): every kernel family with an output store
    cases = [('c3', 2, 24, 32, 64, 64, 3, 1, 1, (0, 1, 3) if dtype == 'f32' else (1, 3, 9, 11, 13, 14, 15, 37)),
             ('c1', 2, 16, 16, 128, 256, 1, 1, 0, (0, 1) if dtype == 'f32' else (1, 7, 9, 11, 36)),
             ('s2', 2, 24, 24, 64, 128, 3, 2, 1, (0, 3) if dtype == 'f32' else (1, 2, 3))]
    for name, B, H, W, Cin, Cout, k, s, p, tiles in cases:
        geo = ops.ConvGeometry(B, H, W, Cin, Cout, k, s, p)
        x = cast(rnd(B, H, W, Cin))
        w = (0.05 * rnd(Cout, k, k, Cin)).to(dev)
        gy = cast(rnd(B, geo.Ho, geo.Wo, Cout))
        for t in tiles:
            try:
                y = ops.conv_fprop(x, w, geo, stats=ops.stats_buffer(Cout, dev), tile=t)
            except RuntimeError:
                continue            # a tile this shape is not offered
            out['%s/%s/fprop/%d' % (dtype, name, t)] = y.float().cpu().numpy()
        out['%s/%s/dgrad' % (dtype, name)] = ops.conv_dgrad(gy, w, geo).float().cpu().numpy()
    # BN passes (16-byte-unit kernels)
    C_ = 64
    x = cast(rnd(4, 20, 24, C_)); res = cast(rnd(4, 20, 24, C_)); gyb = cast(rnd(4, 20, 24, C_))
    st = ops.BNState(C_, dev)
    st.mean.copy_(0.1 * rnd(C_)); st.rstd.copy_(1.0 + 0.1 * rnd(C_).abs())
    st.scale.copy_(st.rstd); st.shift.copy_(-st.mean * st.rstd)
    st.count = x.numel() // C_
    out['%s/bn_apply' % dtype] = ops.bn_apply(x, st).float().cpu().numpy()
    y = ops.bn_apply(x, st, residual=res, want_bits=True)
    out['%s/bn_apply_res' % dtype] = y.float().cpu().numpy()
    gamma = torch.ones(C_, device=dev); gg = torch.zeros(C_, device=dev); gb = torch.zeros(C_, device=dev)
    gx = ops.bn_backward(gyb, y, x, st, gamma, gg, gb)
    out['%s/bn_backward' % dtype] = (gx[0] if isinstance(gx, tuple) else gx).float().cpu().numpy()


if __name__ == '__main__':
    np.savez(sys.argv[1], **outputs())
