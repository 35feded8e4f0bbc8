#!/usr/bin/env python
"""Development tool: the BN / pool passes AS THE STEP CALLS THEM (ops.bn_apply with sign bits, bn_backward single / xmask /
dual-bits, bn_backward_from_sums, the fused stem tail) on configs[2]-sized tensors, solo: time, algorithmic bytes, TB/s.
usage: bn_bench2.py [B H W C]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get('BN_BENCH_LIB'):              # (this tool's own variable: another build of the library, for A/B runs)
    from loans_amd import _lib
    _lib.LIB_PATH = os.path.abspath(os.environ['BN_BENCH_LIB'])
from loans_amd import ops  # noqa: E402


def t(fn, reps=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


B, H, W, C = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (128, 128, 128, 64)))
dt = torch.bfloat16
es = 2
ops.set_compute_dtype('bf16'); ops.set_storage_dtype('bf16')
mk = lambda: torch.randn(B, H, W, C, device='cuda').to(dt)       # noqa: E731
x, x2, g, res = mk(), mk(), mk(), mk()
n = x.numel()


def state(a):
    stats = torch.zeros((ops.STATS_REPLICAS, 2, C), device='cuda', dtype=torch.float64)
    f = a.float().reshape(-1, C).double()
    stats[0, 0], stats[0, 1] = f.sum(0), (f * f).sum(0)
    ones, zeros = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
    return ops.bn_finalize(stats, B * H * W, ones, zeros, zeros.clone(), ones.clone())


st, st2 = state(x), state(x2)
ones = torch.ones(C, device='cuda')
z = lambda: torch.zeros(C, device='cuda')       # noqa: E731
rows = []


def rec(name, ms, nbytes):
    rows.append((name, ms, nbytes))
    print('%-44s %7.3f ms  %6.2f GB  %5.2f TB/s' % (name, ms, nbytes / 1e9, nbytes / ms / 1e9), flush=True)


rec('bn_apply relu (inner, fwd)', t(lambda: ops.bn_apply(x, st, relu=True)), 2 * n * es)
rec('bn_apply relu + residual + bits (outer B)', t(lambda: ops.bn_apply(x, st, relu=True, residual=res, want_bits=True)), 3 * n * es + n // 4)
rec('bn_apply relu + x2 + bits (outer A)', t(lambda: ops.bn_apply(x, st, relu=True, x2=x2, st2=st2, want_bits=True)), 3 * n * es + n // 4)
out = ops.bn_apply(x, st, relu=True, residual=res, want_bits=True)
act = ops.bn_apply(x, st, relu=True)
rec('bn_backward inner two-pass (xmask)', t(lambda: ops.bn_backward(g, act, x, st, ones, z(), z(), mask_is_own_relu=True)), 5 * n * es)
sums = ops.stats_buffer(C, x.device)
rec('bn_backward_from_sums (inner, one pass)', t(lambda: ops.bn_backward_from_sums(g, x, st, sums, ones, z(), z())), 3 * n * es)
rec('bn_backward outer B (bits)', t(lambda: ops.bn_backward(g, out, x, st, ones, z(), z())), 5 * n * es + n // 2)
rec('bn_backward outer A (dual, bits)', t(lambda: ops.bn_backward(g, out, x, st, ones, z(), z(), x2=x2, st2=st2, gamma2=ones, ggamma2=z(), gbeta2=z())),
    8 * n * es + n // 2)
# the stem's tail at configs[2]: conv1 output 128 x 256 x 256 x 64 (2 GB in bf16): a quarter of the batch
Bp, Hp, Cp = 32, 256, 64
xs = torch.randn(Bp, Hp, Hp, Cp, device='cuda').to(dt)
stats = torch.zeros((ops.STATS_REPLICAS, 2, Cp), device='cuda', dtype=torch.float64)
stats[0, 1] = float(Bp * Hp * Hp)
onesp, zerosp = torch.ones(Cp, device='cuda'), torch.zeros(Cp, device='cuda')
sts = ops.bn_finalize(stats, Bp * Hp * Hp, onesp, zerosp, zerosp.clone(), onesp.clone())
y, idx = ops.bn_relu_maxpool(xs, sts)
rec('bn_relu_maxpool (B/4)', t(lambda: ops.bn_relu_maxpool(xs, sts)), xs.numel() * es + y.numel() * (es + 1))
gy = torch.randn_like(y)
rec('pool_bn_backward (reduce + apply, B/4)', t(lambda: ops.pool_bn_backward(gy, idx, xs, sts, onesp, zerosp.clone(), zerosp.clone(), gbias=zerosp.clone())),
    (y.numel() * (es + 1) + xs.numel() * es // 1) + (y.numel() * (es + 1) + 2 * xs.numel() * es))

# the two passes of the stem tail separately (C ABI calls; LOANS_POOL_U16 selects the kernel forms)
lib = ops._lib.load()
B_, H_, C_ = Bp, Hp, Cp
OH = OW = y.shape[1]
sums = torch.zeros((2, C_), device='cuda', dtype=torch.float64)
k = torch.ones((3, C_), device='cuda')
gxs = torch.empty_like(xs)
gb = torch.zeros(C_, device='cuda')
sums_rep = torch.zeros((32, 2, C_), device='cuda', dtype=torch.float64)
rec('  pool_bn_bwd_reduce alone, replicated', t(lambda: ops.check(lib.loans_pool_bn_bwd_reduce_rep_bf16(
    gy.data_ptr(), idx.data_ptr(), xs.data_ptr(), sts.scale.data_ptr(), sts.shift.data_ptr(), sts.mean.data_ptr(), sts.rstd.data_ptr(),
    sums_rep.data_ptr(), 32, B_, H_, H_, C_, OH, OW, ops._stream()), 'r')), y.numel() * (es + 1) + xs.numel() * es)
rec('  pool_bn_bwd_reduce alone', t(lambda: ops.check(lib.loans_pool_bn_bwd_reduce_bf16(
    gy.data_ptr(), idx.data_ptr(), xs.data_ptr(), sts.scale.data_ptr(), sts.shift.data_ptr(), sts.mean.data_ptr(), sts.rstd.data_ptr(),
    sums.data_ptr(), B_, H_, H_, C_, OH, OW, ops._stream()), 'r')), y.numel() * (es + 1) + xs.numel() * es)
reps = torch.zeros((32, C_), device='cuda')
rec('  pool_bn_bwd_apply alone, replicated gxsum', t(lambda: ops.check(lib.loans_pool_bn_bwd_apply_rep_bf16(
    gy.data_ptr(), idx.data_ptr(), xs.data_ptr(), sts.scale.data_ptr(), sts.shift.data_ptr(), k[0].data_ptr(), k[1].data_ptr(),
    k[2].data_ptr(), gxs.data_ptr(), reps.data_ptr(), 32, B_, H_, H_, C_, OH, OW, ops._stream()), 'a')), y.numel() * (es + 1) + 2 * xs.numel() * es)
rec('  pool_bn_bwd_apply alone', t(lambda: ops.check(lib.loans_pool_bn_bwd_apply_bf16(
    gy.data_ptr(), idx.data_ptr(), xs.data_ptr(), sts.scale.data_ptr(), sts.shift.data_ptr(), k[0].data_ptr(), k[1].data_ptr(),
    k[2].data_ptr(), gxs.data_ptr(), gb.data_ptr(), B_, H_, H_, C_, OH, OW, ops._stream()), 'a')), y.numel() * (es + 1) + 2 * xs.numel() * es)
