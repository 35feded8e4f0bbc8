#!/usr/bin/env python
"""Diagnostic: build a -DLOANS_STAMPS copy of the kernel library, run one conv layer and print
the per-wave cycle shares of the K-loop phases (load issue / MFMA / LDS store / barrier)."""
import ctypes, os, subprocess, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, 'loans_amd', 'csrc')
dbg = os.path.join(csrc, 'libloans_hip_stamps.so')
if not os.path.exists(dbg):
    subprocess.check_call('/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DLOANS_STAMPS '
                          '-I%s/include -shared -o %s %s' % (ROOT, dbg, ' '.join('%s/%s' % (csrc, f) for f in (
                              'igemm.hip', 'igemm_bf16.hip', 'halo_bf16.hip', 'stem.hip', 'smalln.hip', 'cropgrad.hip', 'bn_pool.hip', 'misc.hip',
                              'insight.hip', 'resample.hip', 'augment.hip'))), shell=True)
if len(sys.argv) < 11:
    sys.exit(0)         # build only (the GPU box has no reason to compile)
from loans_amd import _lib
_lib.LIB_PATH = dbg
from loans_amd import ops
lib = _lib.load()
lib.loans_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
name, Cin, H, W, Cout, k, s, p = sys.argv[1], *[int(v) for v in sys.argv[2:9]]
B = int(sys.argv[9]); tile = int(sys.argv[10])
geo = ops.ConvGeometry(B, H, W, Cin, Cout, k, s, p)
x = torch.randn(B, H, W, Cin, device='cuda'); w = torch.randn(Cout, k, k, Cin, device='cuda') * 0.05
y = torch.empty(B, geo.Ho, geo.Wo, Cout, device='cuda'); stats = ops.stats_buffer(Cout, 'cuda')
for _ in range(3):
    ops.conv_fprop(x, w, geo, out=y, stats=stats, tile=tile)
torch.cuda.synchronize()
buf = np.zeros(64 * 4 * 8, np.uint64)
assert lib.loans_debug_read_stamps(buf.ctypes.data, buf.size) == 0
st = buf.reshape(64, 4, 8).astype(np.float64)
n = st[..., 6]
per = st[..., :4] / n[..., None]
print('%s tile=%d chunks=%d  per-chunk cycles (median over 256 waves): g0+load=%.0f g1+g2+store=%.0f barrier=%.0f g3=%.0f | loop total/chunk=%.0f | prologue=%.0f loop=%.0f epilogue=%.0f' % (
    name, tile, int(n[0, 0]), *np.median(per.reshape(-1, 4), axis=0), np.median(st[..., 4] / n), np.median(st[..., 7]), np.median(st[..., 4]), np.median(st[..., 5])))
print('  spread of mfma phase per chunk: min %.0f max %.0f' % (per[..., 1].min(), per[..., 1].max()))
