#!/usr/bin/env python
"""Diagnostic: build a -DLOANS_STAMPS copy of the kernel library, run one bf16-storage conv layer and print the per-wave
cycle shares of the K-loop phases of loans_igemm_bf16s (step 0 + DMA, step 1 + DMA, step 2, barrier + DMA wait, step 3).
usage: stamp_run16.py name Cin H W Cout k stride pad B tile"""
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
                              'igemm.hip', 'igemm_bf16.hip', 'halo_bf16.hip', 'wgrad_halo_bf16.hip', 'weightprep.hip', 'stem.hip', 'smalln.hip', 'cropgrad.hip', 'bn_pool.hip', 'misc.hip', 'insight.hip',
                              'resample.hip', 'augment.hip'))), shell=True)
if len(sys.argv) < 11:
    sys.exit(0)         # build only (the GPU box has no reason to compile)
from loans_amd import _lib
_lib.LIB_PATH = dbg
from loans_amd import ops
lib = _lib.load()
lib.loans_debug_read_stamps16.argtypes = [ctypes.c_void_p, ctypes.c_int]
name, Cin, H, W, Cout, k, s, p = sys.argv[1], *[int(v) for v in sys.argv[2:9]]
B = int(sys.argv[9]); tile = int(sys.argv[10])
ops.set_compute_dtype('bf16'); ops.set_storage_dtype('bf16')
geo = ops.ConvGeometry(B, H, W, Cin, Cout, k, s, p)
x = torch.randn(B, H, W, Cin, device='cuda').to(torch.bfloat16); w = torch.randn(Cout, k, k, Cin, device='cuda') * 0.05
for _ in range(3):
    ops.conv_fprop(x, w, geo, tile=tile)
torch.cuda.synchronize()
buf = np.zeros(64 * 4 * 8, np.uint64)
assert lib.loans_debug_read_stamps16(buf.ctypes.data, buf.size) == 0
st = buf.reshape(64, 4, 8).astype(np.float64)
n = st[..., 6]
per = st[..., :5] / n[..., None]
med = np.median(per.reshape(-1, 5), axis=0)
print('%s tile=%d chunks=%d | per-chunk wave cycles (median of 256 waves): step0+dma %.0f  step1+dma %.0f  step2 %.0f  '
      'wait+barrier %.0f  step3 %.0f  | total %.0f  (4 steps x %d MFMAs x 32 cycles = %d pipe cycles per wave)' % (
          name, tile, int(n[0, 0]) + 1, *med, med.sum(), 4, 16 * 32))
bufb = np.zeros(64 * 4 * 4, np.uint64)
lib.loans_debug_read_stamps16b.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.loans_debug_read_stamps16b(bufb.ctypes.data, bufb.size) == 0
sb = bufb.reshape(64, 4, 4).astype(np.float64)
pro, loop, epi = sb[..., 1] - sb[..., 0], sb[..., 2] - sb[..., 1], sb[..., 3] - sb[..., 2]
print('   per block (median of the first 64 tiles): prologue %.0f  K loop %.0f  last chunk + epilogue %.0f cycles; first-wave entries span %.0f cycles' % (
    np.median(pro), np.median(loop), np.median(epi), sb[..., 0].max() - sb[..., 0].min()))
bufc = np.zeros(64 * 4 * 4, np.uint64)
lib.loans_debug_read_stamps16c.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.loans_debug_read_stamps16c(bufc.ctypes.data, bufc.size) == 0
sc = bufc.reshape(64, 4, 4).astype(np.float64)
last, stats, p0, p1 = sc[..., 0] - sb[..., 2], sc[..., 1] - sc[..., 0], sc[..., 2] - sc[..., 1], sc[..., 3] - sc[..., 2]
tail = sb[..., 3] - np.where(sc[..., 3] > 0, sc[..., 3], sc[..., 2])
print('   epilogue (median): last chunk %.0f  statistics %.0f  staging pass 0 + stores %.0f  pass 1 + stores %.0f  after the last pass %.0f cycles' % (
    np.median(last), np.median(stats), np.median(p0), np.median(p1) if (sc[..., 3] > 0).any() else 0, np.median(tail)))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.conv_fprop(x, w, geo, tile=tile)
e1.record(); torch.cuda.synchronize()
print('   launch %.1f us' % (e0.elapsed_time(e1) * 100))
