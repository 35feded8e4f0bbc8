#!/usr/bin/env python
"""Ceilings of this chip, measured: sustained fp32 / bf16 MFMA rate with nothing else in the way, and the HBM fill / copy /
read rates of plain streaming kernels (tools/chip_peaks.hip, built by `make -C tools`).  Prints one line per measurement."""
import ctypes as C, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, 'libchip_peaks.so'))
lib.peak_mfma.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
lib.peak_mem.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(n):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


scratch = torch.zeros(16, device='cuda')
st = torch.cuda.current_stream().cuda_stream
cus = torch.cuda.get_device_properties(0).multi_processor_count
for kind, name, flop in ((0, 'v_mfma_f32_32x32x2_f32', 32 * 32 * 2 * 2), (1, 'v_mfma_f32_32x32x16_bf16', 32 * 32 * 16 * 2)):
    for wps in (1, 2):
        iters = 20000 if kind == 0 else 40000
        ms = timeit(lambda: lib.peak_mfma(kind, wps, iters, scratch.data_ptr(), st))
        total = cus * wps * 4 * iters * 4 * flop
        print('%-26s %d wave(s)/SIMD: %.3f ms  %.1f TFLOP/s' % (name, wps, ms, total / ms / 1e9), flush=True)
nbytes = 2 << 30
a = torch.empty(nbytes // 4, device='cuda'); b = torch.empty(nbytes // 4, device='cuda')
for blocks in (cus * 4, cus * 8, cus * 32):
    for kind, name, moved in ((0, 'fill', 1), (1, 'copy', 2), (2, 'read', 1)):
        ms = timeit(lambda: lib.peak_mem(kind, a.data_ptr(), b.data_ptr() if kind != 2 else scratch.data_ptr(), nbytes, blocks, st))
        print('%-5s 2 GiB, %5d blocks: %.3f ms  %.2f TB/s' % (name, blocks, ms, moved * nbytes / ms / 1e9), flush=True)
ms = timeit(lambda: a.zero_())
print('torch zero_ 2 GiB: %.3f ms  %.2f TB/s' % (ms, nbytes / ms / 1e9))
ms = timeit(lambda: b.copy_(a))
print('torch copy_ 2 GiB: %.3f ms  %.2f TB/s' % (ms, 2 * nbytes / ms / 1e9))
