#!/usr/bin/env python
"""Development tool: achieved HBM bandwidth of the BN / pool passes (fp32 and bf16 storage) on a res2-sized tensor."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
for dt in (torch.float32, torch.bfloat16):
    es = 4 if dt == torch.float32 else 2
    x = torch.randn(B, H, W, C, device='cuda').to(dt)
    g = torch.randn(B, H, W, C, device='cuda').to(dt)
    m = torch.randn(B, H, W, C, device='cuda').to(dt)
    stats = torch.zeros((ops.STATS_REPLICAS, 2, C), device='cuda', dtype=torch.float64)
    stats[0, 0] = 0.0
    stats[0, 1] = float(B * H * W)
    ones, zeros = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
    st = ops.bn_finalize(stats, B * H * W, ones, zeros, zeros.clone(), ones.clone())
    nbytes = x.numel() * es
    ms = t(lambda: ops.bn_apply(x, st, relu=True, residual=m))
    print('%-8s bn_apply+res    %7.3f ms  %6.2f TB/s' % (dt, ms, 3 * nbytes / ms / 1e9))
    gg, gb = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    lib = ops._lib.load()
    sums = torch.zeros((2, C), device='cuda', dtype=torch.float64)
    red = lib.loans_bn_bwd_reduce_bf16 if es == 2 else lib.loans_bn_bwd_reduce_f32
    ms = t(lambda: ops.check(red(g.data_ptr(), m.data_ptr(), x.data_ptr(), st.mean.data_ptr(), st.rstd.data_ptr(), 0, 0, 0,
                                 sums.data_ptr(), B * H * W, C, ops._stream()), 'red'))
    print('%-8s bn_bwd_reduce   %7.3f ms  %6.2f TB/s' % (dt, ms, 3 * nbytes / ms / 1e9))
    k = torch.ones((3, C), device='cuda')
    gx = torch.empty_like(x)
    app = lib.loans_bn_bwd_apply_bf16 if es == 2 else lib.loans_bn_bwd_apply_f32
    ms = t(lambda: ops.check(app(g.data_ptr(), m.data_ptr(), x.data_ptr(), k[0].data_ptr(), k[1].data_ptr(), k[2].data_ptr(),
                                 gx.data_ptr(), 0, 0, 0, 0, 0, B * H * W, C, ops._stream()), 'app'))
    print('%-8s bn_bwd_apply    %7.3f ms  %6.2f TB/s' % (dt, ms, 4 * nbytes / ms / 1e9))

# the stem's pool (conv1 output of config 3: 128 x 256 x 256 x 64 is 2 GB in bf16 -- use a quarter batch)
for dt in (torch.float32, torch.bfloat16):
    es = 4 if dt == torch.float32 else 2
    Bp, Hp, Cp = 32, 256, 64
    x = torch.randn(Bp, Hp, Hp, Cp, device='cuda').to(dt)
    stats = torch.zeros((ops.STATS_REPLICAS, 2, Cp), device='cuda', dtype=torch.float64)
    stats[0, 1] = float(Bp * Hp * Hp)
    ones, zeros = torch.ones(Cp, device='cuda'), torch.zeros(Cp, device='cuda')
    st = ops.bn_finalize(stats, Bp * Hp * Hp, ones, zeros, zeros.clone(), ones.clone())
    y, idx = ops.bn_relu_maxpool(x, st)
    ms = t(lambda: ops.bn_relu_maxpool(x, st))
    nb = x.numel() * es + y.numel() * (es + 1)
    print('%-8s bn_relu_maxpool %7.3f ms  %6.2f TB/s' % (dt, ms, nb / ms / 1e9))
    gy = torch.randn_like(y)
    ms = t(lambda: ops.maxpool_relu_bwd(gy, idx, x, st))
    nb = 2 * x.numel() * es + y.numel() * (es + 1)
    print('%-8s maxpool_relu_bwd %6.3f ms  %6.2f TB/s' % (dt, ms, nb / ms / 1e9))
