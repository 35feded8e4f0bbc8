"""Test helper: Pillow's two integer passes in NumPy, driven by the product's coefficient tables (so that the tables can
be checked on a machine without a GPU)."""
import numpy as np


def two_pass_u8(a, oh, ow, filt='lanczos'):
    from loans_amd.common.datasets.resample import PRECISION_BITS, resample_coeffs
    H, W, _ = a.shape
    hb, hk, _ = resample_coeffs(W, ow, filt)
    vb, vk, _ = resample_coeffs(H, oh, filt)
    half = 1 << (PRECISION_BITS - 1)
    tmp = np.zeros((H, ow, 3), np.uint8)
    for xx in range(ow):
        x0, n = hb[xx]
        acc = half + (a[:, x0:x0 + n, :].astype(np.int64) * hk[xx, :n, None].astype(np.int64)).sum(axis=1)
        tmp[:, xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    out = np.zeros((oh, ow, 3), np.uint8)
    for yy in range(oh):
        y0, n = vb[yy]
        acc = half + (tmp[y0:y0 + n].astype(np.int64) * vk[yy, :n, None, None].astype(np.int64)).sum(axis=0)
        out[yy] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return out
