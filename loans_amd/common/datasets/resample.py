"""The LANCZOS resize of the input contract on the GPU (reference common/datasets/image_dataset.py:16-28: ``resize_image``
= ``PIL.Image.resize((w, h), Image.LANCZOS)``; :98 ``image / 255``).

Pillow's 8-bit resampler is integer arithmetic on per-coordinate coefficient windows (libImaging/Resample.c:
``precompute_coeffs`` in double precision, ``normalize_coeffs_8bpc`` to 22-bit fixed point, a horizontal then a vertical
pass that each round to uint8).  ``lanczos_coeffs`` restates the coefficient computation operation by operation with
``math.sin`` (the same libm call), the two passes are HIP kernels (csrc/resample.hip), and the result is bit-identical to
Pillow (tests/test_resample_cpu.py pins the tables, tests/test_gpu_resample.py the kernels, both against Pillow itself
and against committed vectors).

Decode stays on the host (there is no GPU JPEG decoder in this ROCm image); what moves to the GPU is everything after
it: uint8 frames go up as they are (a quarter of the float bytes), resize + ``/ 255`` + CHW layout happen there."""
import functools
import math

import numpy as np
import torch

from ... import _lib
from ...ops import _ptr, _stream, check

PRECISION_BITS = 32 - 8 - 2
LANCZOS_SUPPORT = 3.0
BILINEAR_SUPPORT = 1.0


def _sinc(x):
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos(x):
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3)
    return 0.0


def _bilinear(x):
    """libImaging/Resample.c bilinear_filter: the triangle on [-1, 1] (``Image.BILINEAR``, the reference's ``Image.LINEAR`` of
    datasets/sheep/paste_and_crop_sheep.py:218 -- the old name of the same filter)"""
    if x < 0.0:
        x = -x
    if x < 1.0:
        return 1.0 - x
    return 0.0


_FILTERS = {'lanczos': (_lanczos, LANCZOS_SUPPORT), 'bilinear': (_bilinear, BILINEAR_SUPPORT)}


def lanczos_coeffs(in_size, out_size):
    return resample_coeffs(in_size, out_size, 'lanczos')


@functools.lru_cache(maxsize=512)
def resample_coeffs(in_size, out_size, filt='lanczos'):
    """(bounds int32 [out_size][2] = (first input index, taps), coefficients int32 [out_size][ksize], ksize) of one axis."""
    filter_fn, filter_support = _FILTERS[filt]
    in0, in1 = 0.0, float(in_size)
    scale = filterscale = (in1 - in0) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = filter_support * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    one = float(1 << PRECISION_BITS)
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        ww = 0.0
        ss = 1.0 / filterscale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [0.0] * xmax
        for x in range(xmax):
            w = filter_fn((x + xmin - center + 0.5) * ss)
            k[x] = w
            ww += w
        for x in range(xmax):
            v = k[x] / ww if ww != 0.0 else k[x]
            kk[xx, x] = int(-0.5 + v * one) if v < 0 else int(0.5 + v * one)
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


_tables = {}


def _device_tables(in_size, out_size, device, filt='lanczos'):
    key = (in_size, out_size, device, filt)
    t = _tables.get(key)
    if t is None:
        b, k, ks = resample_coeffs(in_size, out_size, filt)
        t = _tables[key] = (torch.from_numpy(b).to(device), torch.from_numpy(k).to(device), ks)
    return t


def resize_bilinear(frames_u8, out_hw, as_float=True):
    """Pillow's ``Image.resize(size, Image.BILINEAR)`` (the generator's final resize, paste_and_crop_sheep.py:218)"""
    return resize_lanczos(frames_u8, out_hw, as_float, filt='bilinear')


def resize_lanczos(frames_u8, out_hw, as_float=True, filt='lanczos'):
    """frames_u8: device uint8 tensor [B][H][W][3] (RGB).  Returns [B][3][oh][ow] float32 in [0,1] (= Pillow's LANCZOS
    resize, then ``/ 255``) or, with ``as_float=False``, the resized uint8 [B][oh][ow][3].  ``filt``: 'lanczos' | 'bilinear'
    (the two passes are filter-agnostic: they apply the coefficient tables)."""
    if not (frames_u8.is_cuda and frames_u8.dtype == torch.uint8 and frames_u8.is_contiguous() and frames_u8.dim() == 4
            and frames_u8.shape[3] == 3):
        raise ValueError('frames must be a contiguous device uint8 tensor [B][H][W][3]')
    B, H, W, _ = frames_u8.shape
    oh, ow = int(out_hw[0]), int(out_hw[1])
    dev = frames_u8.device
    lib = _lib.load()
    if (H, W) == (oh, ow):                  # Image.resize returns a copy when the size does not change
        if not as_float:
            return frames_u8.clone()
        out = torch.empty((B, 3, oh, ow), device=dev, dtype=torch.float32)
        check(lib.loans_u8hwc3_to_f32chw(_ptr(frames_u8), _ptr(out), B, H, W, _stream()), 'loans_u8hwc3_to_f32chw')
        return out
    hb, hk, hks = _device_tables(W, ow, dev, filt)
    vb, vk, vks = _device_tables(H, oh, dev, filt)
    tmp = torch.empty((B, H, ow, 3), device=dev, dtype=torch.uint8)
    if as_float:
        out = torch.empty((B, 3, oh, ow), device=dev, dtype=torch.float32)
        fn, name = lib.loans_resize_lanczos_u8_f32, 'loans_resize_lanczos_u8_f32'
    else:
        out = torch.empty((B, oh, ow, 3), device=dev, dtype=torch.uint8)
        fn, name = lib.loans_resize_lanczos_u8, 'loans_resize_lanczos_u8'
    check(fn(_ptr(frames_u8), _ptr(tmp), _ptr(out), B, H, W, oh, ow, _ptr(hb), _ptr(hk), hks, _ptr(vb), _ptr(vk), vks,
             _stream()), name)
    return out


# mirror of loans_resample_job (include/loans_hip.h)
RESAMPLE_JOB = np.dtype([('src_off', '<i8'), ('tmp_off', '<i8'), ('inH', '<i4'), ('inW', '<i4'), ('hb_off', '<i4'), ('hk_off', '<i4'),
                         ('hks', '<i4'), ('vb_off', '<i4'), ('vk_off', '<i4'), ('vks', '<i4'), ('flip', '<i4'), ('_pad', '<i4')])
assert RESAMPLE_JOB.itemsize == 56


class _TableArena:
    """The coefficient tables of every (input size, output size, filter) seen so far, packed into ONE device int32 buffer so
    that a kernel can find any frame's tables by offset (loans_resize_ragged_u8_f32).  The naive crop branch draws a few
    hundred distinct sizes; a table is uploaded the first time its size appears and never again."""

    def __init__(self, device, words=4 << 20):
        self.device, self.buf, self.used, self.where = device, torch.empty(words, device=device, dtype=torch.int32), 0, {}

    def ensure(self, keys):
        """make the tables of every key resident; returns {key: (bounds offset, coefficient offset, ksize)}"""
        tables = {k: resample_coeffs(*k) for k in keys if k not in self.where}
        need = sum(b.size + c.size for b, c, _ in tables.values())
        if self.used + need > self.buf.numel():
            # full: start over with this batch's tables (launches that read the old contents are ahead of the copies below
            # on the same stream)
            tables = {k: resample_coeffs(*k) for k in keys}
            need = sum(b.size + c.size for b, c, _ in tables.values())
            self.used, self.where = 0, {}
            if need > self.buf.numel():
                self.buf = torch.empty(2 * need, device=self.device, dtype=torch.int32)
        for k, (b, c, ks) in tables.items():
            n = b.size + c.size
            self.buf[self.used:self.used + n].copy_(torch.from_numpy(np.concatenate([b.ravel(), c.ravel()])))
            self.where[k] = (self.used, self.used + b.size, ks)
            self.used += n
        return {k: self.where[k] for k in keys}


_arenas = {}


def resize_ragged(dev_all, frames, out_hw, filt='lanczos'):
    """`frames` = [(byte offset in dev_all, H, W[, mirrored])] in batch order, every frame a uint8 [H][W][3] image inside the device buffer
    `dev_all`: Pillow's resize of each to out_hw and `/ 255`, as ONE launch pair -> [N][3][oh][ow] float32 in that order."""
    dev = dev_all.device
    oh, ow = int(out_hw[0]), int(out_hw[1])
    # one arena per (device, stream): a feeding iterator finishes its batches on a stream of its own, and an arena's "full: start
    # over" / regrow are only ordered against the launches of the stream that calls them
    akey = (dev, torch.cuda.current_stream(dev).cuda_stream)
    arena = _arenas.get(akey)
    if arena is None:
        arena = _arenas[akey] = _TableArena(dev)
    frames = [tuple(f) + (False,) * (4 - len(f)) for f in frames]
    at = arena.ensure({(W, ow, filt) for _, _, W, _ in frames} | {(H, oh, filt) for _, H, _, _ in frames})
    jobs = np.zeros(len(frames), RESAMPLE_JOB)
    tmp_off = 0
    for j, (off, H, W, mirrored) in enumerate(frames):
        jobs[j] = (off, tmp_off, H, W) + at[(W, ow, filt)] + at[(H, oh, filt)] + (int(bool(mirrored)), 0)
        tmp_off += (H * ow * 3 + 15) & ~15
    ring, slot, host = _staging.get(jobs.nbytes, kind='jobs')
    host[:jobs.nbytes].numpy()[:] = jobs.view(np.uint8)
    jobs_d = host[:jobs.nbytes].to(dev, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    ring['events'][slot] = ev
    tmp = torch.empty(tmp_off, device=dev, dtype=torch.uint8)
    out = torch.empty((len(frames), 3, oh, ow), device=dev, dtype=torch.float32)
    check(_lib.load().loans_resize_ragged_u8_f32(_ptr(dev_all), _ptr(tmp), _ptr(out), _ptr(jobs_d), len(frames), _ptr(arena.buf),
                                                 max(f[1] for f in frames), oh, ow, _stream()), 'loans_resize_ragged_u8_f32')
    return out


class _Staging:
    """Pinned host buffers for the uint8 upload, re-used from batch to batch (``pin_memory()`` of a fresh 200 MB tensor per
    batch cost more than the upload itself).  Two buffers per calling thread take turns; each carries the event of the last
    copy that read it, and is only overwritten once that copy has finished."""

    def __init__(self):
        import threading
        self.slots, self.lock = {}, threading.Lock()       # (the dict is shared by every feeding thread: guarded; a ring is its thread's own)

    def get(self, nbytes, kind='frames'):
        import threading
        key = (threading.get_ident(), kind)
        with self.lock:
            ring = self.slots.setdefault(key, {'i': 0, 'bufs': [None, None], 'events': [None, None]})
        i = ring['i'] = (ring['i'] + 1) % 2
        if ring['events'][i] is not None:
            ring['events'][i].synchronize()
        buf = ring['bufs'][i]
        if buf is None or buf.numel() < nbytes:
            buf = ring['bufs'][i] = torch.empty(max(int(nbytes * 1.25), 1 << 20), dtype=torch.uint8).pin_memory()
        return ring, i, buf

    def drop(self, thread_ident):
        """forget (and free) the buffers of a thread that is gone -- a feed iterator's finish thread"""
        with self.lock:
            gone = [self.slots.pop(k) for k in [k for k in self.slots if k[0] == thread_ident]]
        for ring in gone:
            for ev in ring['events']:
                if ev is not None:
                    ev.synchronize()


_staging = _Staging()


def release_staging(thread_ident):
    _staging.drop(thread_ident)


def upload_frames(images, device, map_fn=map):
    """uint8 HWC RGB arrays of ONE size -> a device tensor [N][H][W][3]: staged into the thread's pinned ring (``map_fn`` spreads
    the copies over a pool), one asynchronous upload"""
    device = torch.device(device)
    H, W = images[0].shape[:2]
    per = H * W * 3
    ring, slot, host = _staging.get(per * len(images))
    host_np = host.numpy()

    def stage(k):
        np.copyto(host_np[k * per:(k + 1) * per].reshape(H, W, 3), images[k])

    list(map_fn(stage, range(len(images))))
    dev_all = host[:per * len(images)].to(device, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    ring['events'][slot] = ev
    return dev_all.view(len(images), H, W, 3)


def frames_to_device(images, out_hw, device, augment_rows=None, map_fn=map):
    """``[ImageDataset.get_example(i) for i in batch]`` for decoded frames: a list of uint8 HWC RGB arrays of any sizes
    -> one device batch [N][3][oh][ow] float32 in [0,1], in input order.  Frames of equal size are uploaded (pinned,
    asynchronous, ONE copy for all groups) and resized together; ``map_fn`` (a thread pool's ``map``) spreads the copies into
    the staging buffer."""
    device = torch.device(device)
    groups = {}
    for i, im in enumerate(images):
        if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
            raise ValueError('frames must be uint8 HWC RGB arrays')
        groups.setdefault(im.shape[:2], []).append(i)
    jobs, off, spans = [], 0, []
    for (H, W), idx in groups.items():
        off = (off + 255) & ~255                # every size group starts on a 256-byte boundary
        spans.append(((H, W), idx, off))
        for i in idx:
            jobs.append((i, off, H, W))
            off += H * W * 3
    total = off
    ring, slot, host = _staging.get(total)
    host_np = host.numpy()

    # a horizontally flipped view (random_flip of the naive branch: a negative stride along W) would be copied 3 bytes at a
    # time; it is staged as it lies in memory -- whole rows -- and mirrored by the resampling kernel (ragged path only)
    ragged = augment_rows is None and len(groups) > 1
    mirrored = [ragged and im.strides[1] < 0 for im in images]

    def stage(job):
        i, o, H, W = job
        np.copyto(host_np[o:o + H * W * 3].reshape(H, W, 3), images[i][:, ::-1] if mirrored[i] else images[i])

    list(map_fn(stage, jobs))
    dev_all = host[:total].to(device, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    ring['events'][slot] = ev
    if augment_rows is None and len(spans) > 1:
        # many sizes (the naive crop branch: up to one size per frame): ONE launch pair for the whole batch, in input order.
        # Per size group this used to be two launches, three allocations and an index upload that synchronised the stream --
        # 128 groups per batch ran the GPU stages at 2 700 frames/s (profiles/r3_feed_bench.txt)
        where = {}
        for (H, W), idx, o in spans:
            for n, i in enumerate(idx):
                where[i] = (o + n * H * W * 3, H, W, mirrored[i])
        return resize_ragged(dev_all, [where[i] for i in range(len(images))], out_hw)
    out = None
    for (H, W), idx, o in spans:
        frames = dev_all[o:o + len(idx) * H * W * 3].view(len(idx), H, W, 3)
        if augment_rows is not None:            # the imgaug branch (augment.py), per image, before the resize (reference :80-93)
            from .augment import apply_device
            frames = apply_device(frames, [augment_rows[i] for i in idx])
        batch = resize_lanczos(frames, out_hw)
        if len(spans) == 1:
            return batch                        # one size: already in input order
        if out is None:
            out = torch.empty((len(images), 3, int(out_hw[0]), int(out_hw[1])), device=device, dtype=torch.float32)
        out[torch.as_tensor(idx, device=device)] = batch
    return out
